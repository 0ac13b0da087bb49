// VLQ line select, one WORKGROUP per query (round 4).  Same selection as line_select_kernel (line.hip) -- the w1 smallest
// (key, candidate index) of the nprobe x nedge candidate lines, key = t > 0 ? b2 : b2 - 0.25 t^2 / c2 (sumAlongRowsWithOrder2,
// gpu/impl/BroadcastSum.cu:477-560), emitted in ascending order (:538-553) -- and the same outputs, bit for bit.
//
// line_select_kernel keeps a running 1024-key selection in ONE wave: with w1 = 1024 of 4096 candidates every 64-candidate trip
// is a flush (sort 64 + merge 1024), and 2000 queries are 2000 waves on 256 CUs: 0.54 ms at the reference driver's geometry, a
// sixth of the whole search once the scan stopped reading far-end rows (line16c.hip).  Here:
//   1. 256 threads compute all keys at once (every gather of the query's distance row in flight together), ordered 32-bit
//      images in LDS;
//   2. the w1-th smallest is found by an MSB-first radix select on the 32-bit image (<= 4 passes of 256-bin LDS histograms);
//      keys equal to it are admitted in candidate order (a block-wide prefix count), which is the (key, index) order;
//   3. the <= 1024 winners are sorted once (bitonic sort in LDS, 64-bit keys image << 32 | index);
//   4. lengths -> scan positions by a block scan in emitted order; the compact records for the scan kernels are placed by a
//      counting sort over the anchors' probe ranks (lines sharing an anchor adjacent, anchors in probe order).  The order of an
//      anchor's lines among themselves is not the old kernel's (atomic slots): no scan kernel depends on it -- scan positions
//      and ranks travel in the records.
#include <cstdio>

#include "line.h"
#include "wave_topk.cuh"

namespace vlq {

#ifdef VLQ_LS2_TIMING
__device__ unsigned long long g_ls2_t[8];
#define LS2_T(i) if (t == 0 && (blockIdx.x % 31) == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_ls2_t[i], now_ - tlast_); tlast_ = now_; }
#else
#define LS2_T(i)
#endif

namespace {

constexpr int NT = 256;

// inclusive block scan of one uint32 per thread (256 threads), result for this thread; tot = block total
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t* wsum /* [4] LDS */, int t, uint32_t* tot) {
    const int lane = t & 63, wave = t >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const uint32_t o = __shfl_up(incl, sft, 64);
        if (lane >= sft) incl += o;
    }
    __syncthreads();                       // wsum may still be read from a previous scan
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t s = wsum[w]; if (w < wave) base += s; total += s; }
    *tot = total;
    return base + incl;
}

}  // namespace

// KP = candidates per thread (contiguous: thread t owns candidates t*KP .. t*KP+KP-1), NS = sort size (power of two >= w1)
template <int KP, int NS>
__global__ __launch_bounds__(NT) void line_select2_kernel(
    const float* __restrict__ dist, int64_t nq, int nlist, const int64_t* __restrict__ keys, int nprobe,
    const int32_t* __restrict__ edge_info, const float* __restrict__ edge_dist, int nedge, int w1,
    int32_t* __restrict__ sel_line, float* __restrict__ sel_b2, float* __restrict__ sel_g,
    const int64_t* __restrict__ line_off, const int64_t* __restrict__ line_len, int max_line_codes,
    LineMeta* __restrict__ sel_meta, int32_t* __restrict__ sel_cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    uint32_t* kimg = reinterpret_cast<uint32_t*>(smraw);                 // [KP][NT] ordered key images (0xffffffff: invalid); thread t owns column t
    u64* win = reinterpret_cast<u64*>(kimg + NT * KP);                   // [NS] winners, then sorted
    uint32_t* hist = reinterpret_cast<uint32_t*>(win + NS);              // [256] radix bins; later [nprobe] anchor bins
    uint32_t* misc = hist + (nprobe > 256 ? ((nprobe + 1) & ~1) : 256);  // [16]: scan scratch, digit, counts (even bin count: ebits below is 8-byte aligned)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t q = blockIdx.x;
    const float* row = dist + q * nlist;
    const int64_t* kq = keys + q * nprobe;
    const int num = nprobe * nedge;

#ifdef VLQ_LS2_TIMING
    unsigned long long tlast_ = wall_clock64();
#endif
    // ---- 1. keys ----  Staged and branch-free: all anchors, then all edge entries, then all row gathers of a chunk of 16
    // candidates are requested before the first is used (written as one loop with `if (valid)` bodies the compiler kept each
    // candidate's three dependent loads in sequence: 100 us of the 220 a workgroup took).
    uint32_t nvalid_t = 0;
    constexpr int CH = KP < 16 ? KP : 16;
#pragma unroll 1
    for (int r0 = 0; r0 < KP; r0 += CH) {
        const int i0 = t * KP + r0;
        int64_t cr[CH];
        int er[CH];
        bool ok[CH];
        {
            int pr = i0 / nedge, e = i0 - pr * nedge;
#pragma unroll
            for (int r = 0; r < CH; r++) {
                ok[r] = i0 + r < num;
                cr[r] = kq[ok[r] ? pr : 0];
                er[r] = e;
                if (++e == nedge) { e = 0; pr++; }
            }
        }
        int sr[CH];
        float c2r[CH];
#pragma unroll
        for (int r = 0; r < CH; r++) {
            ok[r] = ok[r] && cr[r] >= 0;
            if (!ok[r]) { cr[r] = 0; er[r] = 0; }
            sr[r] = edge_info[cr[r] * nedge + er[r]];
            c2r[r] = edge_dist[cr[r] * nedge + er[r]];
        }
        float a2r[CH], b2r[CH];
#pragma unroll
        for (int r = 0; r < CH; r++) { a2r[r] = row[sr[r]]; b2r[r] = row[cr[r]]; }
#pragma unroll
        for (int r = 0; r < CH; r++) {
            const float g = __fsub_rn(a2r[r], b2r[r]);
            const float tt = __fsub_rn(g, c2r[r]);
            // BroadcastSum.cu:503-507: beyond the near end -> distance to c, else to the line
            const float key = (tt > 0.f) ? b2r[r] : __fsub_rn(b2r[r], __fdiv_rn(__fmul_rn(__fmul_rn(0.25f, tt), tt), c2r[r]));
            // (the wave select never admits FLT_MAX or NaN: `key < thr` with thr <= FLT_MAX; same rule here)
            const bool keep = ok[r] && key < 3.402823466e+38f;
            kimg[(r0 + r) * NT + t] = keep ? f32_to_ordered(key) : 0xffffffffu;
            nvalid_t += keep ? 1u : 0u;
        }
    }
    uint32_t nvalid;
    (void)block_scan_incl(nvalid_t, misc, t, &nvalid);
    LS2_T(0)
    const uint32_t want = min((uint32_t)w1, nvalid);                      // winners

    // ---- 2. radix select of the want-th smallest image (want >= 1) ----
    uint32_t prefix = 0, need = want;            // need-th smallest among the images matching `prefix` on the decided bits
    int decided = 0;                             // bits decided (from the top)
    if (want > 0 && want < nvalid) {
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[t] = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < KP; r++) {
                const uint32_t img = kimg[r * NT + t];
                const bool match = img != 0xffffffffu && (decided == 0 || (img >> (32 - decided)) == (prefix >> (32 - decided)));
                if (match) atomicAdd(&hist[(img >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (wave == 0) {                     // bin holding the need-th element: 4 bins per lane
                const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                uint32_t incl = h0 + h1 + h2 + h3;
                const uint32_t mine = incl;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) {
                    const uint32_t o = __shfl_up(incl, sft, 64);
                    if (lane >= sft) incl += o;
                }
                const uint32_t excl = incl - mine;
                if (excl < need && need <= incl) {           // exactly one lane
                    uint32_t below = excl, b = 4 * lane;
                    if (need > below + h0) { below += h0; b++; if (need > below + h1) { below += h1; b++; if (need > below + h2) { below += h2; b++; } } }
                    misc[8] = b;
                    misc[9] = need - below;                   // rank inside the bin
                    misc[10] = hist[b];
                }
            }
            __syncthreads();
            prefix |= misc[8] << shift;
            need = misc[9];
            decided += 8;
            if (misc[10] == need) break;         // every image of the bin is a winner: no finer cut needed
        }
    } else {
        decided = 0;                             // all valid images win (prefix unused)
    }
    LS2_T(1)
    // threshold: images strictly below (on the decided bits) win; images equal on the decided bits: all of them if the bin
    // was taken whole, else (32 bits decided, images EQUAL to the threshold) the `need` first in candidate order
    const bool all_win = !(want > 0 && want < nvalid);
    const int dshift = 32 - decided;
    auto cls = [&](uint32_t img) -> int {        // 0 below, 1 on the threshold, 2 above / invalid
        if (img == 0xffffffffu) return 2;
        if (all_win) return 0;
        const uint32_t a = dshift >= 32 ? 0u : (img >> dshift), b = dshift >= 32 ? 0u : (prefix >> dshift);
        return a < b ? 0 : (a == b ? 1 : 2);
    };
    uint32_t eq_t = 0;
#pragma unroll
    for (int r = 0; r < KP; r++) eq_t += cls(kimg[r * NT + t]) == 1 ? 1u : 0u;
    uint32_t eq_tot;
    const uint32_t eq_before = block_scan_incl(eq_t, misc, t, &eq_tot) - eq_t;
    // ---- 3. winners into LDS (any order), then one bitonic sort of 64-bit (image, index) keys ----
    for (int i = t; i < NS; i += NT) win[i] = kMaxKey;
    if (t == 0) misc[12] = 0;
    __syncthreads();
    {
        uint32_t eqc = eq_before;
#pragma unroll
        for (int r = 0; r < KP; r++) {
            const uint32_t img = kimg[r * NT + t];
            const int c = cls(img);
            bool take = c == 0;
            if (c == 1) { take = all_win || eqc < need; eqc++; }
            if (take) {
                const uint32_t slot = atomicAdd(&misc[12], 1u);
                if (slot < (uint32_t)NS) win[slot] = ((u64)img << 32) | (uint32_t)(t * KP + r);
            }
        }
    }
    __syncthreads();
    LS2_T(2)
    for (int size = 2; size <= NS; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = t; p < NS / 2; p += NT) {
                const int lo = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
                const int hi = lo | stride;
                const bool up = (lo & size) == 0 || size == NS;
                const u64 a = win[lo], b = win[hi];
                if ((a > b) == up) { win[lo] = b; win[hi] = a; }
            }
            __syncthreads();
        }
    }
    LS2_T(3)
    // ---- 4. outputs.  Thread t owns emitted places w = t*WP .. (contiguous: the scan of the lengths is in emitted order).
    // Loads staged like the keys': winners -> anchors -> edge entries and line extents -> row values.
    constexpr int WP = NS / NT > 0 ? NS / NT : 1;
    uint32_t len_r[WP], any_r[WP];
    int64_t c_r[WP], off_r[WP];
    int pr_r[WP], s_r[WP];
    int32_t line_r[WP];
    float c2_r[WP], b2_r[WP], g_r[WP];
    bool ok_r[WP];
#pragma unroll
    for (int r = 0; r < WP; r++) {
        const int w = t * WP + r;
        const u64 k64 = (w < w1 && w < NS) ? win[w] : kMaxKey;
        ok_r[r] = k64 != kMaxKey;
        const int i = ok_r[r] ? (int)(uint32_t)k64 : 0;
        pr_r[r] = i / nedge;
        const int e = i - pr_r[r] * nedge;
        c_r[r] = kq[pr_r[r]];
        if (!ok_r[r] || c_r[r] < 0) c_r[r] = 0;            // (a winner's anchor is never negative)
        line_r[r] = (int32_t)(c_r[r] * nedge + e);
    }
#pragma unroll
    for (int r = 0; r < WP; r++) {
        s_r[r] = edge_info[line_r[r]];
        c2_r[r] = edge_dist[line_r[r]];
        int64_t l64 = 0;
        off_r[r] = 0;
        if (sel_meta) {
            off_r[r] = line_off[line_r[r]];
            l64 = line_len ? line_len[line_r[r]] : line_off[line_r[r] + 1] - off_r[r];
            if (l64 > max_line_codes) l64 = max_line_codes;
        }
        len_r[r] = ok_r[r] ? (uint32_t)l64 : 0u;
        any_r[r] = len_r[r] > 0 ? 1u : 0u;
    }
#pragma unroll
    for (int r = 0; r < WP; r++) {
        b2_r[r] = row[c_r[r]];
        g_r[r] = __fsub_rn(row[s_r[r]], b2_r[r]);
    }
    uint32_t lsum = 0, nsum = 0;
#pragma unroll
    for (int r = 0; r < WP; r++) {
        const int w = t * WP + r;
        if (w < w1 && w < NS) {
            sel_line[q * w1 + w] = ok_r[r] ? line_r[r] : -1;
            sel_b2[q * w1 + w] = ok_r[r] ? b2_r[r] : 0.f;
            sel_g[q * w1 + w] = ok_r[r] ? g_r[r] : 0.f;
        }
        lsum += len_r[r];
        nsum += any_r[r];
    }
    for (int w = NS + t; w < w1; w += NT) { sel_line[q * w1 + w] = -1; sel_b2[q * w1 + w] = 0.f; sel_g[q * w1 + w] = 0.f; }
    LS2_T(4)
    if (!sel_meta) return;
    uint32_t ltot, ntot;
    uint32_t pos = block_scan_incl(lsum, misc, t, &ltot) - lsum;         // scan position of this thread's first line
    uint32_t rnk = block_scan_incl(nsum, misc, t, &ntot) - nsum;         // its rank among the non-empty kept lines
    // counting sort of the non-empty winners by probe rank (anchor): bins in LDS.  nedge <= 64: a bin also keeps the bitmap of
    // its winning edges, so a line's place inside its anchor's group is its edge's rank -- ascending line id = ascending
    // address, the order of the one-wave kernel; more edges than a 64-bit word: atomic slots (any order is correct)
    u64* ebits = reinterpret_cast<u64*>(misc + 16);                       // [nprobe]
    const bool bitmaps = nedge <= 64;
    for (int i = t; i < nprobe; i += NT) { hist[i] = 0; ebits[i] = 0; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < WP; r++)
        if (any_r[r]) {
            atomicAdd(&hist[pr_r[r]], 1u);
            if (bitmaps) atomicOr(reinterpret_cast<unsigned long long*>(&ebits[pr_r[r]]), 1ull << (line_r[r] - (int32_t)(c_r[r] * nedge)));
        }
    __syncthreads();
    // exclusive prefix over the nprobe bins (<= 1024): each thread owns ceil(nprobe / 256) consecutive bins
    {
        const int per = (nprobe + NT - 1) / NT;
        uint32_t sm = 0;
        for (int i = 0; i < per; i++) { const int bb = t * per + i; if (bb < nprobe) sm += hist[bb]; }
        uint32_t tot;
        uint32_t run = block_scan_incl(sm, misc, t, &tot) - sm;
        for (int i = 0; i < per; i++) {
            const int bb = t * per + i;
            if (bb < nprobe) { const uint32_t c = hist[bb]; hist[bb] = run; run += c; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < WP; r++) {
        if (any_r[r]) {
            const int e = line_r[r] - (int32_t)(c_r[r] * nedge);
            const uint32_t slot = bitmaps ? hist[pr_r[r]] + (uint32_t)__popcll(ebits[pr_r[r]] & ((1ull << e) - 1ull))
                                          : atomicAdd(&hist[pr_r[r]], 1u);
            LineMeta m;
            m.off = off_r[r];
            m.len = (int32_t)len_r[r];
            m.line = line_r[r];
            m.s = s_r[r];
            m.c2 = c2_r[r];
            m.b2 = b2_r[r];
            m.g = g_r[r];
            m.pos0 = pos;
            m.rank = (int32_t)rnk;
            m.anchor = (int32_t)c_r[r];
            m.pad1 = 0;
            sel_meta[q * w1 + slot] = m;
        }
        pos += len_r[r];
        rnk += any_r[r];
    }
    LS2_T(5)
    if (t == 0) sel_cnt[q] = (int32_t)ntot;
#ifdef VLQ_LS2_TIMING
    if (t == 0 && (blockIdx.x % 31) == 0) atomicAdd(&g_ls2_t[7], 1ull);
#endif
}

template <int KP, int NS>
static void launch_ls2(const float* dist, int64_t nq, int nlist, const int64_t* keys, int nprobe, const int32_t* edge_info,
                       const float* edge_dist, int nedge, int w1, int32_t* sel_line, float* sel_b2, float* sel_g, hipStream_t s,
                       const int64_t* line_off, const int64_t* line_len, int max_line_codes, LineMeta* sel_meta, int32_t* sel_cnt) {
    const size_t smem = (size_t)NT * KP * 4 + (size_t)NS * 8 + (size_t)(nprobe > 256 ? ((nprobe + 1) & ~1) : 256) * 4 + 16 * 4 + (size_t)nprobe * 8;
    ensure_dynamic_lds(reinterpret_cast<const void*>(line_select2_kernel<KP, NS>), smem);
    hipLaunchKernelGGL((line_select2_kernel<KP, NS>), dim3((unsigned)nq), dim3(NT), smem, s, dist, nq, nlist, keys, nprobe, edge_info,
                       edge_dist, nedge, w1, sel_line, sel_b2, sel_g, line_off, line_len, max_line_codes, sel_meta, sel_cnt);
}

// shapes the workgroup kernel serves: up to 16 384 candidates (64 per thread), w1 <= 1024
bool line_select2_supports(int nprobe, int nedge, int w1) {
    const int64_t num = (int64_t)nprobe * nedge;
    return num >= 512 && num <= 16384 && w1 >= 1 && w1 <= 1024 && nprobe <= 1024;
}

void launch_line_select2(const float* dist, int64_t nq, int nlist, const int64_t* keys, int nprobe, const int32_t* edge_info,
                         const float* edge_dist, int nedge, int w1, int32_t* sel_line, float* sel_b2, float* sel_g, hipStream_t s,
                         const int64_t* line_off, const int64_t* line_len, int max_line_codes, LineMeta* sel_meta, int32_t* sel_cnt) {
    if (nq <= 0) return;
    const int num = nprobe * nedge;
    const int kp = (num + NT - 1) / NT;
#define VLQ_LS2(KP, NS) launch_ls2<KP, NS>(dist, nq, nlist, keys, nprobe, edge_info, edge_dist, nedge, w1, sel_line, sel_b2, sel_g, s, \
                                           line_off, line_len, max_line_codes, sel_meta, sel_cnt)
#define VLQ_LS2_NS(KP)                          \
    do {                                        \
        if (w1 <= 256) VLQ_LS2(KP, 256);        \
        else VLQ_LS2(KP, 1024);                 \
    } while (0)
    if (kp <= 4) VLQ_LS2_NS(4);
    else if (kp <= 16) VLQ_LS2_NS(16);
    else VLQ_LS2_NS(64);
#undef VLQ_LS2_NS
#undef VLQ_LS2
#ifdef VLQ_LS2_TIMING
    {
        unsigned long long h[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ls2_t), sizeof(h));
        if (h[7]) fprintf(stderr, "[ls2 timing] keys %.1f radix %.1f compact %.1f sort %.1f out1 %.1f out2 %.1f us per workgroup (%llu)\n", h[0] * 0.01 / h[7], h[1] * 0.01 / h[7], h[2] * 0.01 / h[7], h[3] * 0.01 / h[7], h[4] * 0.01 / h[7], h[5] * 0.01 / h[7], h[7]);
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ls2_t), z, sizeof(z));
    }
#endif
}

}  // namespace vlq
