// Host-side launch interface of the HIP kernels (kernels.hip).  Everything takes
// raw device pointers and a stream; no allocation, no synchronisation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlq {

// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) to `bytes` on
// the CURRENT device.  The attribute is per device and the library may serve several devices
// (and host threads) of one process, so the high-water mark is kept per (kernel, device).
void ensure_dynamic_lds(const void* kernel, size_t bytes);
// loads the code objects of the search path's kernels (the runtime loads a translation unit's code object at the first use of one
// of its kernels: ~0.2-0.5 ms each); called with the lists, so that a caller's first search does not pay for it
void preload_search_kernels();
void preload_scan16_kernels();       // scan16.hip
void preload_coarse_screen_kernels();  // coarse_screen.hip
void preload_scanm_kernels();        // scanm.hip

// row norms in the reference's SSE order (utils.cpp:538-556, :675-682)
void launch_row_norms(const float* x, int64_t n, int d, float* out, hipStream_t s);

// out[i][j] = (qn[i] + cn[j]) - 2 * <q_i, c_j>   (utils.cpp:884), inner product =
// k-ordered f32 MFMA chain.  out is [nq][nlist].  qn == nullptr (d <= 128 only, coarse_norms_fused_ok): the
// kernel computes |q_i|^2 itself, in fvec_norm_L2sqr's order, from the query tile it stages anyway.
// tmin != nullptr (only if coarse_tile_minima_ok): also tmin[i][t] = min of out[i][64t .. 64t+63]
// out_rows: rows the caller allocated for `out`; with room for whole 128-row blocks (>= round_up(nq, 128)) the
// software-pipelined loop runs, which stores its blocks without a row guard (rows past nq: written, never read)
void launch_coarse_distances(const float* q, const float* c, const float* qn, const float* cn,
                             float* out, int64_t nq, int nlist, int d, hipStream_t s, float* tmin = nullptr,
                             int64_t out_rows = 0);
inline bool coarse_norms_fused_ok(int d) { return d <= 128; }
// wide rows and few probes: the distance kernel can hand the select a [nq][nlist/64] matrix of tile minima
bool coarse_tile_minima_ok(int nlist, int d, int nprobe);
// 1-NN (the assignment of add / encode): out == nullptr and tmin = [nq][nlist / 64] 64-bit keys -- the
// distance kernel writes no matrix, launch_coarse_argmin reduces the keys to (distance, centroid)
bool coarse_argmin_ok(int nlist, int d);
void launch_coarse_argmin(const void* tile_keys, int64_t nq, int nlist, float* cdis, int64_t* keys, hipStream_t s);

// float16-screened coarse stage (coarse_screen.hip): approximate distance matrix from half copies, rigorous keep test,
// exact fp32 distances (the f32 MFMA kernel's fmaf chain) for the kept columns only -- the keys, distances and tie order
// of the matrix path.  approx: [roundup128(nq)][nlist] floats of scratch; kept_total (optional): += kept columns.
bool coarse_screen_shape_ok(int nlist, int d, int nprobe);
// one pass over x [n][d]: half(scale * (x - mu)) in the MFMA operand order ([roundup128(n)] rows x roundup16(d) halves), per row
// |x|^2 (reference order), |x - mu|^2 and the half-range flag of scale * (x - mu) (flags optional)
void launch_screen_prep(const float* x, const float* mu, int64_t n, int d, float scale, void* out_half, float* norms, float* norms_c,
                        unsigned char* flags, hipStream_t s);
// nprobe == 1 (the assignment of add / encode): tile minima only (tmin_ws [nq][nlist / 64] floats), no matrix
bool coarse_screen_nn_shape_ok(int nlist, int d);
void launch_coarse_screened_nn(const float* q, const void* q_half, const unsigned char* q_flags, const float* c, const void* c_half,
                               const float* qn, const float* cn, const float* qn_c, const float* cn_c, float* tmin_ws, int64_t nq, int nlist,
                               int d, float scale, float cmax, float cmax0, float* cdis, int64_t* keys, unsigned int* exact_rows,
                               hipStream_t s);
size_t coarse_screen_keep_bytes(int64_t nq, int nlist);      // keep_ws of launch_coarse_screened
bool coarse_screen_matrix_free_ok(int nlist, int nprobe);    // the screen without the half matrix (tmin_ws then always needed)
// The scan order's histogram (launch_query_order: counting sort of the queries by their nearest centroid) taken along by the
// coarse stage's last kernel, which holds every row's nearest centroid anyway: hist[bin(keys[q][0])] += 1 (hist zeroed by the
// caller; the bins are launch_query_order's: query_order_bins).  One launch and its gap less per search.
struct OrderHist { int* hist = nullptr; const int* list_rank = nullptr; int shift = 0, nbins = 0, nlist = 0; };
void query_order_bins(int nlist, int* shift, int* nbins);
// qn / cn: exact squared norms (reference order) of queries / centroids; qn_c / cn_c: of the centred ones; cmax = max |c - mu|
void launch_coarse_screened(const float* q, const void* q_half, const unsigned char* q_flags, const float* c, const void* c_half,
                            const float* qn, const float* cn, const float* qn_c, const float* cn_c, float* approx,
                            float* tmin_ws /* [nq][nlist / 64] floats when nlist > 8192, else unused */, void* keep_ws,
                            int64_t nq, int nlist, int d, int nprobe, float scale, float cmax, float cmax0 /* max |c| */, float* cdis,
                            int64_t* keys,
                            unsigned long long* kept_total, unsigned int* exact_rows /* += rows the screen could not decide (optional) */,
                            hipStream_t s, OrderHist oh = OrderHist(), bool* hist_done = nullptr /* set when the histogram was taken */);

// filtered coarse stage (no distance matrix; kernels.hip): a sample of the column tiles (stride s) gives every
// row an exact upper bound of its nprobe-th smallest distance, the full pass keeps only the elements at or
// below it as (distance, column) keys, the select runs over those (rows whose buffer overflowed are redone)
bool coarse_filter_ok(int nlist, int d, int nprobe, int64_t nq, int* stride_out, int* cap_out);
void launch_sample_tiles(const float* c, const float* cn, int nlist, int d, int s, float* cs, float* cns, hipStream_t st);
void launch_coarse_distances_filtered(const float* q, const float* c, const float* qn, const float* cn, int64_t nq, int nlist,
                                      int d, const float* bound, int64_t bound_stride, unsigned long long* cand,
                                      unsigned char* cnt, hipStream_t s);
void launch_coarse_select_cand(const unsigned long long* cand, const unsigned char* cnt, int64_t nq, int nprobe, float* cdis,
                               int64_t* keys, const float* q, const float* c, const float* qn, const float* cn, int nlist, int d,
                               hipStream_t s);

// < 20 queries: direct fvec_L2sqr per pair (utils.cpp:757-786)
void launch_coarse_distances_direct(const float* q, const float* c, float* out, int64_t nq,
                                    int nlist, int d, hipStream_t s);

// per row: the nprobe smallest (distance, column), ascending; pads -1 / FLT_MAX
// tmin != nullptr: two-level select that reads only the tiles whose minimum can matter
void launch_coarse_select(const float* dist, int64_t nq, int nlist, int nprobe, float* cdis,
                          int64_t* keys, hipStream_t s, const float* tmin = nullptr);

// PQ tables.  mode 0: <x_m, cent_mj>  (ProductQuantizer.cpp:424-436)
//             mode 1: |x_m - cent_mj|^2 (ProductQuantizer.cpp:410-422)
//             mode 2: rnorm[m][j] + 2 <x_m, cent_mj>  (IndexIVFPQ.cpp:423-429)
// x [nv][d], cent [M][ksub][dsub], out [nv][M][ksub]
void launch_pq_tables(const float* x, int64_t nv, int d, const float* cent, int M, int ksub,
                      int dsub, const float* rnorm, int mode, float* out, hipStream_t s);

// one visited probe of a query as the second build of the list-owned schedule reads it (scan16o.hip)
struct OwnRec {
    int32_t key;      // list id
    uint32_t len;     // codes in the list
    int64_t off;      // first code of the list
    float dis0;       // coarse distance
    uint32_t pos0;    // scan position of the list's first code (prefix over the query's probes in coarse order)
};
static_assert(sizeof(OwnRec) == 24, "OwnRec is copied as six dwords");

struct ScanArgs {
    const uint8_t* codes;        // [ntotal][M] list-contiguous
    const int64_t* ids;          // [ntotal]
    const int64_t* list_off;     // [nlist+1] list starts
    const int64_t* list_len = nullptr;   // [nlist] lengths, or nullptr: packed lists (len = off[i+1] - off[i])
    const float* term2;          // [nlist][M*ksub]   (table mode 1) or nullptr
    const float* qtab;           // [nq][M*ksub] per-query table (ip table or distance table)
    const float* queries;        // [nq][d]          (table mode 0 only)
    const float* coarse;         // [nlist][d]       (table mode 0 only)
    const float* pq_cent;        // [M][ksub][dsub]  (table mode 0 only)
    const float* pq_cent_t = nullptr;  // [M][dsub][ksub] transposed copy (scan16 fused tables)
    const int64_t* keys;         // [nq][nprobe]
    const float* coarse_dis;     // [nq][nprobe]
    float* D;                    // [nq][k]
    int64_t* I;                  // [nq][k]
    unsigned long long* ncode;   // accumulated number of visited codes
    int* bad_key;                // set to 1 if a key >= nlist was met
    int64_t nq;
    int nprobe, k, M, ksub, dsub, d, nlist;
    int table_mode;              // 0: by_residual, no table; 1: by_residual + term2; 2: not by_residual
    int64_t max_codes;
    int store_pairs;
    int imi_nbits = 0;           // > 0: table mode 2 -- key = i0 | i1 << imi_nbits, term2 rows per coarse SUB-index
    const int* qorder = nullptr; // optional processing order of the queries (scan16 only)
    int nsplit = 1;              // scan16: workgroups per query; > 1: D / I are [nsplit][nq][k] partial rows
    int long_lists = 0;          // scan16: mean list length >= 4 chunks -- selects the pipelined pair loop for k > 64 too
    // scan16, split TAIL of a batch that fills the chip a fractional number of times (nsplit == 1): on every XCD the last
    // tail_r queries of its chunk are scanned by tail_p workgroups each (contiguous ranges of the walking order, like
    // nsplit) that write partial rows [tail_p][8 * tail_r][k] to tail_D / tail_I and their query to tail_rows; the whole
    // queries in front of them write D / I directly.  launch_merge_topk(..., tail_rows) joins the parts.
    int tail_r = 0, tail_p = 1;
    float* tail_D = nullptr;
    int64_t* tail_I = nullptr;
    int* tail_rows = nullptr;    // [8 * tail_r], preset to -1
    int xcd_chunk = 0;           // set by the launcher
    const int* walk_flag = nullptr;  // optional: walk_stat_kernel's 32 counts of probes shared by neighbours of the scan order ...
    int walk_clock = 0;              // A/B: fixed period of the walk clock in 10 ns ticks (VLQ_WALK_CLOCK)
    int* walk_state = nullptr;       // per XCD (16 ints apart): running mean of a workgroup's walk time, kept across launches
    int walk_scale = 1000;           // period = measured walk time x this / 1000
    int walk_limit = 0;              // ... list-id order only while their sum is <= this
    int short_keep_order = 0;    // scan16_short: walk a query's multi-index cells in coarse order (A/B: VLQ_SHORT_KEEP_ORDER) instead of by halves
    int walk_first = -1;         // walk_order.cuh: < 0 = probes in coarse-distance order, else this many nearest first, the rest by list id
    int grid_per_xcd = 0;        // set by the launcher: workgroups per XCD (xcd_chunk unless the tail is split)
    // list-owned schedule (scan16 only, DESIGN.md "list-owned schedule"): the lists are cut into 8
    // partitions of neighbouring lists, one per XCD; a workgroup serves the probes of ONE query that fall
    // into ONE partition and leaves its k best raw keys (ordered distance << 32 | scan position) in
    // part_keys [nq][8][k]; owned_merge joins a query's parts in the global (distance, position) order
    const uint8_t* list_part = nullptr;      // [nlist] partition 0..7 of every list
    const int* own_order = nullptr;          // [8][nq] the queries with probes in partition x, scheduling order
    const int* own_count = nullptr;          // [8]
    unsigned long long* part_keys = nullptr; // [nq][8][k]
    const uint8_t* part_mask = nullptr;      // [nq] bit x: the query has a probe in partition x
    int qtab_scaled = 0;                     // qtab already holds (-2) * <q_m, cent_mj>
    // second build (scan16o.hip): per-probe records grouped by partition [nq][nprobe], item entries [8][nq] = (query, first
    // record | count << 16)
    const OwnRec* own_recs = nullptr;
    const uint2* own_items = nullptr;
    // float16 look-up tables (scan16h.hip): half(term2) [nlist][M*ksub] and half(-2 <q_m, cent_mj>) [nq][M*ksub]
    const uint16_t* term2h = nullptr;
    const uint16_t* qtabh = nullptr;
};
void launch_scan(const ScanArgs& a, hipStream_t s);
// 8-, 32- and 64-byte codes (M x 8 bit), table mode 1 / table type 2, per-query table in a.qtab: scan16's organisation over the
// code size (scanm.hip); same results as launch_scan
bool scanm_supports(const ScanArgs& a);
bool scanm0_supports(const ScanArgs& a);      // table mode 0, 8- / 16-byte codes (launch_scanm serves it too)
void launch_scanm(const ScanArgs& a, hipStream_t s);
// specialisation for M = 16, ksub = 256, table_mode = 1 (scan16.hip)
void launch_scan16(const ScanArgs& a, hipStream_t s);
const char* last_scan16_shape();      // "scan16_kernel<KPL, NW, NBUF, PIPE, IMI, OWNED>" of this thread's last launch
// list-owned schedule of the same kernel: launch_owned_order prepares own_order / own_count / part_mask,
// launch_qtab16 the per-query table (-2 <q_m, cent_mj>, [nq][16][256]), launch_scan16_owned scans the
// (query, partition) items and launch_owned_merge writes the final rows
void launch_owned_order(const int64_t* keys, int64_t nq, int nprobe, int nlist, const int* list_rank,
                        const uint8_t* list_part, int* hist /* [2][8][nlist] + 8 */, int* minr /* [nq][8] */,
                        int* own_order, int* own_count, uint8_t* part_mask, hipStream_t s);
inline size_t owned_hist_ints(int nlist) { return (size_t)16 * nlist + 64; }
void launch_qtab16(const float* queries, int64_t nq, const float* pq_cent_t, float* qtab, hipStream_t s);
void launch_scan16_owned(const ScanArgs& a, hipStream_t s);
void launch_owned_merge(const ScanArgs& a, hipStream_t s);
// second build of the list-owned schedule (scan16o.hip): launch_owned2_prepare writes the per-probe records, the items of
// every partition, own_count and part_mask (hist: [2][8][nlist] ints of scratch, minr [nq][8], seg [nq][8]);
// launch_scan16_owned2 scans the items (nbuf = 1 / 2 table buffers); launch_owned_merge joins the parts as before
bool scan16o_supports(const ScanArgs& a);
void launch_owned2_prepare(const ScanArgs& a, const int* list_rank, int* hist, int* minr, uint32_t* seg, uint2* items,
                           int* own_count, uint8_t* part_mask, OwnRec* recs, hipStream_t s);
void launch_scan16_owned2(const ScanArgs& a, int nbuf, hipStream_t s);
// same shape with float16 look-up tables (useFloat16LookupTables; scan16h.hip), k <= 256
bool scan16h_supports(const ScanArgs& a);
void launch_scan16h(const ScanArgs& a, hipStream_t s);
// *out = bit pattern of the largest |x[i]| (the half-range check of the float16 tables)
void launch_max_abs(const float* x, int64_t n, unsigned int* out, hipStream_t s);
// same shape, 256 < k <= 1024: one selection per workgroup instead of one per wave (scan16k.hip)
void launch_scan16_bigk(const ScanArgs& a, hipStream_t s);
// same shape, indexes with a few codes per list (multi-index): no per-probe LUT (scan16.hip)
void launch_scan16_short(const ScanArgs& a, hipStream_t s);
// the same organisation for the other code sizes (4 ... 64 bytes in the steps scanm.hip serves; per-query table in a.qtab)
bool scanm_short_supports(const ScanArgs& a);
void launch_scanm_short(const ScanArgs& a, hipStream_t s);
// counting sort of query ids by nearest coarse centroid: hist [nlist+1] ints scratch
// ints of scratch launch_query_order needs in `hist`: 2 x this
inline size_t query_order_bins_padded(int nlist) {
    int shift = 0;
    while (((int64_t)nlist >> shift) > 16384) shift++;
    const size_t nbins = (size_t)((((int64_t)nlist - 1) >> shift) + 2);
    return (nbins + 63) & ~(size_t)63;
}
// samples neighbour pairs of the scan order and decides the walking order of the probes (scan16.hip, walk_order.cuh)
// what a launch without measured walk times seeds its clock period from (walk_order.cuh): the list lengths of the sampled probes
// and the workgroups that will run side by side
struct WalkSeed { const int64_t* list_off = nullptr; const int64_t* list_len = nullptr; int nlist = 0; int slots = 0; };
int launch_walk_stat(const int64_t* keys, const int* qorder, int64_t nq, int nprobe, int* part, int* walk_state, hipStream_t s,
                     WalkSeed seed = WalkSeed());
// list_rank (optional): bins are the spatial ranks of the lists instead of the list ids
void launch_query_order(const int64_t* keys, int64_t nq, int nprobe, int nlist, int* hist,
                        int* qorder, hipStream_t s, const int* list_rank = nullptr, int* walk_part = nullptr, int* walk_state = nullptr,
                        WalkSeed seed = WalkSeed(), bool run_walk_stat = true, bool hist_ready = false);
// walk_part (optional): the walking-order statistic (32 counts, see launch_walk_stat) is computed along with the order when
// run_walk_stat (otherwise the counts of an earlier search stay and the placement kernel freezes this launch's clock period)
int walk_stat_samples(int64_t nq, int nprobe);

// merge of per-shard results [nparts][nq][k] into the global top-k (list-sharded multi-GPU mode)
void launch_merge_topk(const float* Dp, const int64_t* Ip, int64_t nq, int k, int nparts, float* D,
                       int64_t* I, hipStream_t s, const int* row_map = nullptr);

// out[i][0..dc) = x[i][col0 .. col0+dc)
void launch_gather_cols(const float* x, int64_t n, int d, int col0, int dc, float* out, hipStream_t s);

// MultiIndexQuantizer::search for 2 sub-quantizers (IndexPQ.cpp:804-857): sorted[m] =
// the T smallest entries of the m-th distance table, ascending, as (value, index);
// replays MinSumK (IndexPQ.cpp:690-778) per query.  heap_* = scratch [nq][2*k].
void launch_imi_minsum(const float* sv0, const int64_t* si0, const float* sv1, const int64_t* si1, int T,
                       int64_t nq, int k, int kc, int imi_nbits, float* heap_val, int64_t* heap_id,
                       float* sums, int64_t* keys, hipStream_t s);

// imi_wide.hip: the T <= 4096 smallest entries of every row in (value, column) order (ncol % 4 == 0; ld = row stride),
// and the MinSumK replay for 64 < k <= 4096 by one wave per query with its heap in LDS
bool row_select_sorted_ok(int ncol, int T);
void launch_row_select_sorted(const float* dist, int64_t nq, int64_t ld, int ncol, int T, float* sv, int64_t* si, hipStream_t s);
bool imi_minsum_wide_ok(int T, int k, int kc);
void launch_imi_minsum_wide(const float* sv0, const int64_t* si0, const float* sv1, const int64_t* si1, int T, int64_t nq, int k,
                            int kc, int imi_nbits, float* sums, int64_t* keys, hipStream_t s);

// out[m][c][j] = in[m][j][c]
void launch_transpose_pq(const float* in, int M, int ksub, int dsub, float* out, hipStream_t s);

// encode path (IndexIVFPQ.cpp:192-231, ProductQuantizer.cpp:311-336)
// imi_nbits > 0: `coarse` is the IMI codebook [2][2^imi_nbits][d/2] and the centroid of key
// is the concatenation of its two sub-centroids (MultiIndexQuantizer::reconstruct)
void launch_residual_encode(const float* x, int64_t n, int d, const float* coarse,
                            const int64_t* assign, int by_residual, const float* cent, int M,
                            int ksub, int dsub, uint8_t* codes, hipStream_t s, int imi_nbits = 0);

}  // namespace vlq
