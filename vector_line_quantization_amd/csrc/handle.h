// Internal: the index handle and small host helpers shared by api.hip and line_api.hip.
#pragma once
#include "../../include/vlq_ivfpq.h"
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace vlq_detail {

inline std::string& err_slot() { static thread_local std::string s; return s; }

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err_slot() = buf;
    return code;
}

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(VLQ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                             \
    } while (0)

#define TRY(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != VLQ_OK) return rc_; \
    } while (0)

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return VLQ_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e != hipSuccess) { p = nullptr; return fail(VLQ_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        cap = want;
        return VLQ_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// 0 = ordinary (pageable) host memory, 1 = device / managed memory, 2 = page-locked host memory that kernels can
// address (hipHostMalloc / hipHostRegister: what GpuResources::getPinnedMemory hands out); *dev = its device-side address
inline int ptr_kind(const void* p, void** dev = nullptr) {
    if (!p) return 0;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) return 1;
    if (attr.type == hipMemoryTypeHost && attr.devicePointer) { if (dev) *dev = attr.devicePointer; return 2; }
    return 0;
}

inline bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

}  // namespace vlq_detail
using namespace vlq_detail;

struct AppendWs {   // lists.h AppendWorkspace (kept opaque here: handle.h is included by lists.h)
    DevBuf cnt, cstart, keys_in, keys_out, sort_tmp;
};

struct vlq_ivfpq_s {
    int device = 0, d = 0, nlist = 0, M = 0, nbits = 0, ksub = 0, dsub = 0;
    int by_residual = 1, use_precomputed_table = 1;
    int64_t max_codes = 0;
    int64_t ntotal = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;

    // inverted lists: list i at [list_off[i], list_off[i] + list_len[i]), capacity list_off[i+1] - list_off[i]
    DevBuf coarse, cnorm, pq, pq_t, rnorm, term2, codes, ids, list_off, list_len;
    DevBuf list_rank;             // [nlist] int: spatial order of the lists (query scheduling only)
    DevBuf list_part;             // [nlist] u8: partition 0..7 of neighbouring lists, one per XCD (list-owned schedule)
    bool have_rank = false;
    // scan schedule of the 16-byte kernel: 0 = automatic (= 1 today), 1 = one workgroup per query (query-major),
    // 2 = list-owned (one workgroup per (query, list partition)), 3 = its second build (scan16o.hip; 4: with two
    // table buffers).  Speed only, never results.
    int scan_schedule = 0;
    DevBuf ws_own_hist, ws_own_minr, ws_own_order, ws_own_count, ws_part_mask, ws_part_keys, ws_own_recs, ws_own_seg, ws_own_items;
    // filtered coarse stage: sampled column tiles of the centroid matrix (stride coarse_s_stride; 0 = none yet),
    // candidate keys and counts
    DevBuf coarse_s, cnorm_s, ws_cand, ws_cnt;
    int coarse_s_stride = 0;
    // float16 screen of the coarse stage (coarse_screen.hip): half(scale * centroids) [nlist][roundup16(d)], the power-of-two
    // scale, the largest centroid norm; 0 = off, 1 = on (default where the shape allows)
    // per centroid set (the flat quantizer; each half of a multi-index): half copy in operand order, the centroids' mean,
    // centred squared norms, the power-of-two scale, max |c - mu|, max |c|
    struct ScreenSet { DevBuf half, mu, norm_c; float scale = 1.f, cmax = 0.f, cmax0 = 0.f; bool ok = false; };
    ScreenSet screen, imi_screen[2];
    DevBuf ws_qn_c, ws_xh, ws_xflags, ws_kept, ws_screen_cnt;
    // second half of a multi-index coarse stage, screened beside the first on an auxiliary stream (imi_page): its own copies
    // of the per-half workspaces, the stream, fork / join events
    struct HalfWs { DevBuf xh, xflags, qn, qn_c, cand, tmin; } imi_ws2;
    hipStream_t imi_stream = nullptr;
    hipEvent_t imi_fork = nullptr, imi_join = nullptr;
    // rows the screen could not decide (too many / too few columns kept -> done exactly, slowly): counted on the device, mirrored
    // into page-locked memory by an asynchronous copy after every batch and looked at before the next -- an index whose data
    // defeat the bound (0.5 % of the rows) goes back to the matrix path for good
    unsigned int* screen_cnt_host = nullptr;
    uint64_t screen_rows_seen = 0, screen_rows_copied = 0;
    int coarse_screen = 1;
    int coarse_filter = 0;        // 1 (VLQ_COARSE_FILTER=1): the filtered coarse stage -- exact, measured SLOWER than the
                                  // matrix path (0.218 against 0.159 ms at C1), kept for A/B only
    // MultiIndexQuantizer coarse quantizer (2 x imi_nbits): codebook [2][kc][d/2], its norms,
    // and the kc virtual full vectors whose term2 rows make table type 2
    int imi_nbits = 0;
    DevBuf imi_cent, imi_norm, imi_virtual, ws_imi;
    bool have_coarse = false, have_pq = false, term2_valid = false, have_lists = false;
    std::vector<int64_t> h_list_off, h_list_len;   // host copies, refreshed on demand (lists_sync_host)
    bool h_lists_stale = false;
    AppendWs ws_append;

    // workspace
    DevBuf ws_Dp, ws_Ip;          // partial top-k rows of the split scan (small batches)
    DevBuf ws_Dr, ws_Ir;          // rows of the runs of a search with more than VLQ_MAX_NPROBE probes
    DevBuf ws_keys_run, ws_cdis_run;   // ... and one run's keys / coarse distances (ws_keys / ws_cdis hold the whole probe list)
    DevBuf ws_x, ws_qn, ws_dist, ws_keys, ws_cdis, ws_qtab, ws_D, ws_I, ws_misc, ws_keys_in,
        ws_cdis_in, ws_codes, ws_assign, ws_hist, ws_qorder, ws_tmin, walk_state;
    int64_t walk_key = -1;       // (nprobe, k, batch class) the walk times in walk_state were measured for
    vlq::OrderHist order_hist;   // the scan order's histogram taken along by the coarse stage (vlq_ivfpq_search only)
    bool order_hist_ready = false;
    int64_t walk_stat_calls = 0; // searches of that key so far (the neighbour-sharing statistic is re-sampled on some of them only)
    // what the last search_dev scan launch was (vlq_ivfpq_last_scan_info): kernel shape, the walking-order rule and the
    // device-side statistic it was decided from (32 counts behind the scan order)
    char last_scan[64] = "";
    int last_walk_first = -1, last_walk_limit = 0, last_walk_samples = 0;
    bool last_walk_counts = false;   // walk_counts holds the 32 counts of the last launch's walk statistic
    DevBuf walk_counts;
    // float16 look-up tables of the plain IVFPQ scan (vlq_ivfpq_set_float16_tables): half(term2), per-page half(term3)
    bool fp16_tables = false, term2h_valid = false;
    DevBuf term2h, ws_qtabh;
    DevBuf stats;   // [0] ncode (u64), [1] bad key flag (int)
    uint64_t stat_nq = 0;

    // profiling
    bool prof = false, prof_scan_only = false;
    int prof_every = 1;              // scan-only timing of every prof_every-th search call (profile mode 3: every 4th)
    uint64_t prof_seq = 0;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    struct Pending { hipEvent_t a, b; int stage; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[3] = {0, 0, 0};
    int64_t prof_calls = 0;
};


// internal entry points implemented in api.hip
namespace vlq_detail {
int set_dev(vlq_ivfpq_t h);
int stage_in(vlq_ivfpq_t h, const void* src, size_t bytes, DevBuf& ws, const void** out);
int stage_out(void* dst, size_t bytes, DevBuf& ws, void** dev, bool* need_copy, bool* zero_copy = nullptr);
int finish_outputs(vlq_ivfpq_t h, bool copyD, void* D, const void* Dd, size_t bytesD, bool copyI,
                   void* I, const void* Id, size_t bytesI);
int ensure_term2(vlq_ivfpq_t h);
int64_t query_page(vlq_ivfpq_t h);
// coarse stage of ONE page (n <= query_page); keep_matrix: the [n][nlist] distance matrix must be
// left in h->ws_dist (the VLQ line select reads it) -- a 1-NN assignment otherwise never writes it;
// zero_qnorm drops |q|^2 (the VLQ path, impl/Distance.cu:286-291)
int coarse_page(vlq_ivfpq_t h, int64_t n, const float* x_dev, int nprobe, float* cdis_dev,
                int64_t* keys_dev, bool zero_qnorm, bool direct, bool keep_matrix = false);
}  // namespace vlq_detail
