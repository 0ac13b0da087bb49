// Multi-index coarse stage beyond 64 probes (round 6): the reference's MultiIndexQuantizer::search has no limit on k
// (IndexPQ.cpp:804-857) and its own drivers ask for 2048 cells per query (tests/sift1b_imi_pq.cpp:363,
// tests/deep1b_imi_pq.cpp, tests/deep1b16_imi_pq.cpp).  Two kernels:
//
//   row_select_sorted_kernel   the T smallest entries of a distance-table row in (value, column) order, T <= 4096 --
//                              what SemiSortedArray's first T ranks are up to exactly equal values (IndexPQ.cpp:524-607;
//                              oracle/ivfpq_oracle.cpp:356-360 orders equal values by column).  One workgroup per row:
//                              radix select of the T-th smallest ordered key on (key - row minimum), 11 bits a pass,
//                              stopping as soon as the boundary bin is taken whole (two passes on real data), wave-aggregated
//                              compaction, one bitonic sort of the T 64-bit (value, column) keys in LDS.  The running
//                              wave selection of kernels.hip (WaveSelect<16>) stops at 1024.
//   imi_minsum_wide_kernel     the MinSumK replay (IndexPQ.cpp:690-778) for 64 < k <= 4096: one wave per query, its binary
//                              heap ({sum : term}, 8 bytes an entry) in LDS.  The heap never holds more than k entries (two
//                              after the first cell, one more per emitted cell, the last cell's pushes feed nothing and are
//                              skipped), so k = 2048 is 16 KB: ten queries per CU.  The walk is one chain of dependent steps,
//                              and a single wave issues an instruction every four cycles at best, so the wave's lanes do a
//                              sift's comparisons side by side: a sift-down reads five levels of descendants in one LDS
//                              round trip (62 lanes, one entry each), every pair picks its smaller child at once, a lane is on
//                              the path when all its ancestors in the fan were picked (two ballots), and the entries on the
//                              path move up in one write; a push reads all ancestors of the new slot at once and moves the ones the new value
//                              beats.  Same comparisons on the same heap positions as Heap.h:89-127 with CMin => the same
//                              pops in the same order, ties and twice-emitted cells included.  The output holds TERMS during
//                              the walk and is turned into keys by all 64 lanes behind a barrier; a cell's four table
//                              entries are fetched by four lanes (vector loads: a scalar load in flight would turn every
//                              LDS wait into a wait for memory) before the sift-down and used after it.  Measured, 10 000
//                              queries, 2 x 14 bits, k = 2048: lane 0 walking alone with heap and tables in LDS (five
//                              queries per CU, ~800 scalar instructions per cell) 47 ms; this kernel: see DESIGN.md 3.1.
//                              (The thread-per-query kernels of kernels.hip keep 32 heaps per workgroup in LDS up to k = 128
//                              and fall back to a heap in global memory beyond.)
#include "kernels.h"
#include "wave_topk.cuh"

namespace vlq {

namespace {
constexpr int kSelThreads = 256;
constexpr int kSelBits = 11, kSelBins = 1 << kSelBits;

// exclusive prefix of one value per thread over the 256 threads; total in *tot (LDS scratch of 8 words)
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* wsum, uint32_t* tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_scan_incl_u32(v);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < kSelThreads / 64; w++) base += (w < wave) ? wsum[w] : 0u;
    if (tot) *tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + incl - v;
}
}  // namespace

template <int TP>       // sort size: 2048 or 4096 (T <= TP)
__global__ __launch_bounds__(kSelThreads) void row_select_sorted_kernel(const float* __restrict__ dist, int64_t ld, int ncol, int T,
                                                                        float* __restrict__ sv, int64_t* __restrict__ si) {
    __shared__ u64 buf[TP];
    __shared__ uint32_t hist[kSelBins];
    __shared__ uint32_t wsum[8];
    __shared__ uint32_t s_min, s_max, s_bin, s_need, s_cnt, s_out;
    const int t = threadIdx.x, lane = t & 63;
    const int64_t q = blockIdx.x;
    const float4* row4 = reinterpret_cast<const float4*>(dist + q * ld);
    const int n4 = ncol >> 2;
    if (t == 0) { s_min = 0xFFFFFFFFu; s_max = 0u; s_out = 0u; }
    __syncthreads();
    // row minimum and maximum of the ordered keys
    {
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
        for (int j = t; j < n4; j += kSelThreads) {
            const float4 v = row4[j];
            const uint32_t a = f32_to_ordered(v.x), b = f32_to_ordered(v.y), c = f32_to_ordered(v.z), d = f32_to_ordered(v.w);
            mn = min(min(mn, min(a, b)), min(c, d));
            mx = max(max(mx, max(a, b)), max(c, d));
        }
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { mn = min(mn, lane_xor_u32(mn, s)); mx = max(mx, lane_xor_u32(mx, s)); }
        if (lane == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
    }
    __syncthreads();
    const uint32_t kmin = s_min, span = s_max - kmin;
    // radix select on rel = key - kmin, most significant bits first: an element is below the boundary when
    // (rel >> shift) < P, on it when == P
    int shift = span ? max(0, 32 - (int)__clz(span) - kSelBits) : 0;
    uint32_t P = 0;
    uint32_t need = (uint32_t)T;       // how many of the boundary elements belong to the T smallest
    bool first = true, whole = false;  // whole: the boundary bin is taken entirely
    int prev_shift = 32;
    while (true) {
        for (int b = t; b < kSelBins; b += kSelThreads) hist[b] = 0;
        __syncthreads();
        const uint32_t dmask = first ? 0xFFFFFFFFu : ((1u << (prev_shift - shift)) - 1u);
        for (int j = t; j < n4; j += kSelThreads) {
            const float4 v = row4[j];
            const uint32_t r[4] = {f32_to_ordered(v.x) - kmin, f32_to_ordered(v.y) - kmin, f32_to_ordered(v.z) - kmin, f32_to_ordered(v.w) - kmin};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bool on = first || (prev_shift >= 32 ? true : ((r[c] >> prev_shift) == P));
                if (on) atomicAdd(&hist[(r[c] >> shift) & dmask], 1u);
            }
        }
        __syncthreads();
        // bin holding the need-th boundary element: 8 consecutive bins per thread
        uint32_t loc[kSelBins / kSelThreads], sum = 0;
#pragma unroll
        for (int i = 0; i < kSelBins / kSelThreads; i++) { loc[i] = hist[t * (kSelBins / kSelThreads) + i]; sum += loc[i]; }
        uint32_t run = block_scan_excl(sum, wsum, nullptr);
#pragma unroll
        for (int i = 0; i < kSelBins / kSelThreads; i++) {
            if (run < need && need <= run + loc[i]) { s_bin = (uint32_t)(t * (kSelBins / kSelThreads) + i); s_need = need - run; s_cnt = loc[i]; }
            run += loc[i];
        }
        __syncthreads();
        const uint32_t b = s_bin;
        P = first ? b : ((P << (prev_shift - shift)) | b);
        need = s_need;
        whole = s_cnt == need;
        first = false;
        if (whole || shift == 0) break;
        prev_shift = shift;
        shift = max(0, shift - kSelBits);
    }
    __syncthreads();
    // compaction: everything below the boundary, and the boundary bin when it is taken whole
    for (int j0 = 0; j0 < n4; j0 += kSelThreads) {
        const int j = j0 + t;
        uint32_t keyv[4] = {0, 0, 0, 0};
        uint32_t cnt = 0;
        bool take[4] = {false, false, false, false};
        if (j < n4) {
            const float4 v = row4[j];
            keyv[0] = f32_to_ordered(v.x); keyv[1] = f32_to_ordered(v.y); keyv[2] = f32_to_ordered(v.z); keyv[3] = f32_to_ordered(v.w);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t hi = (keyv[c] - kmin) >> shift;
                take[c] = hi < P || (whole && hi == P);
                cnt += take[c] ? 1u : 0u;
            }
        }
        const uint32_t incl = wave_scan_incl_u32(cnt);
        uint32_t base = 0;
        if (lane == 63 && incl) base = atomicAdd(&s_out, incl);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
        uint32_t pos = base + incl - cnt;
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (take[c]) buf[pos++] = ((u64)keyv[c] << 32) | (uint32_t)(4 * j + c);
    }
    __syncthreads();
    if (!whole) {
        // exactly equal values straddle the T-th rank: the first `need` of them in column order (one wave walks the row)
        if (t < 64) {
            const uint32_t thr = kmin + P;         // shift == 0 here
            uint32_t got = 0, out = s_out;
            for (int j0 = 0; j0 < ncol && got < need; j0 += 64) {
                const int j = j0 + lane;
                const bool eq = j < ncol && f32_to_ordered(dist[q * ld + j]) == thr;
                const u64 m = __ballot(eq);
                const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (eq && got + before < need) buf[out + got + before] = ((u64)thr << 32) | (uint32_t)j;
                got += (uint32_t)__popcll(m);
            }
        }
        __syncthreads();
    }
    for (int e = T + t; e < TP; e += kSelThreads) buf[e] = kMaxKey;
    __syncthreads();
    // ascending bitonic sort of TP keys
    for (int size = 2; size <= TP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < TP / 2; i += kSelThreads) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const u64 a = buf[lo], c = buf[hi];
                const bool up = (lo & size) == 0;
                if ((a > c) == up) { buf[lo] = c; buf[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int e = t; e < T; e += kSelThreads) {
        const u64 key = buf[e];
        sv[q * T + e] = ordered_to_f32((uint32_t)(key >> 32));
        si[q * T + e] = (int64_t)(uint32_t)key;
    }
}

bool row_select_sorted_ok(int ncol, int T) { return T >= 1 && T <= 4096 && T <= ncol && (ncol & 3) == 0; }

void launch_row_select_sorted(const float* dist, int64_t nq, int64_t ld, int ncol, int T, float* sv, int64_t* si, hipStream_t s) {
    if (nq <= 0) return;
    if (T <= 2048)
        hipLaunchKernelGGL(row_select_sorted_kernel<2048>, dim3((unsigned)nq), dim3(kSelThreads), 0, s, dist, ld, ncol, T, sv, si);
    else
        hipLaunchKernelGGL(row_select_sorted_kernel<4096>, dim3((unsigned)nq), dim3(kSelThreads), 0, s, dist, ld, ncol, T, sv, si);
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void imi_minsum_wide_kernel(const float* __restrict__ sv0, const int64_t* __restrict__ si0,
                                                             const float* __restrict__ sv1, const int64_t* __restrict__ si1, int T,
                                                             int64_t nq, int k, int kc, int imi_nbits, float* __restrict__ sums,
                                                             int64_t* keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    // heap slot i (1-based, Heap.h's indexing) at byte 8 * (i - 1); {sum bits : term}, term = r0 | r1 << 16
    u64* heap = reinterpret_cast<u64*>(smraw) - 1;
    const float* t0 = sv0 + q * T;
    const float* t1 = sv1 + q * T;
    float* out_s = sums + q * k;
    int64_t* out_k = keys + q * k;
    auto fval = [](u64 e) { return __uint_as_float((uint32_t)(e >> 32)); };
    auto entry = [](float v, int id) { return ((u64)__float_as_uint(v) << 32) | (uint32_t)id; };
    auto rl = [](int x, int l) { return __builtin_amdgcn_readlane(x, l); };
    // A lane's place in the fan of descendants a sift-down reads at once: lanes 0-1 the children of the current node, 2-5 its
    // grandchildren, ... 30-61 the fifth level; lane = 2^L - 2 + p, slot = (node << L) + p.  Siblings are lane ^ 1.
    const int fan_t = lane + 2;
    const int L = 31 - __clz(fan_t);
    const int fp = fan_t - (1 << L);
    const bool fan = L <= 5;
    // the lanes of this lane's ancestors inside the fan, and itself: it lies on the sift's path iff all of them were picked
    u64 anc = fan ? (1ull << lane) : ~0ull;
    for (int l = L - 1, pp = fp >> 1; fan && l >= 1; l--, pp >>= 1) anc |= 1ull << ((1 << l) - 2 + pp);
    const bool even = (lane & 1) == 0;
    u64 top = 0;                                        // heap[1], carried in registers

    // Heap.h:110-127 with CMin; n = size after the push.  Lane j holds the j-th ancestor of the new slot; the new value climbs
    // past the nearest ancestors it beats (consecutive ones from the slot up), each of which moves one slot down its path.
    auto push = [&](int n, float val, int id) {
        const int a = lane < 30 ? (n >> (lane + 1)) : 0;
        const u64 e = a >= 1 ? heap[a] : 0ull;
        const u64 climb = __ballot(a >= 1 && val < fval(e));
        const int c = __builtin_ctzll(~climb);
        if (lane < c) heap[n >> lane] = e;
        const int pos = n >> c;
        const u64 ne = entry(val, id);
        if (lane == 0) heap[pos] = ne;
        if (pos == 1) top = ne;
    };
    // Heap.h:89-108; n = size before the pop.  Five levels of descendants in one LDS round trip; every pair picks its smaller
    // child at once (the reference's rule, incl. `i2 == k + 1`), the picks are followed down from the node while the last entry
    // does not win, the picked entries on that path move up one level, all in one write.
    auto pop = [&](int n) {
        const u64 last = heap[n];
        const float val = fval(last);
        int i = 1;
        while (2 * i <= n) {
            const int slot = (i << L) + fp;
            const bool valid = fan && slot <= n;
            const u64 e = valid ? heap[slot] : 0ull;
            const float v = fval(e);
            const float vs = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xF, 0xF, true));
            // first child of the pair picked: its sibling is slot n + 1, or it is the smaller one
            const bool first = even ? (slot + 1 > n || v < vs) : (slot > n || vs < v);
            const bool chosen = valid && (even == first);
            const u64 picked = __ballot(chosen);
            const u64 on = __ballot((picked & anc) == anc);
            const u64 stop = __ballot(val < v) & on;
            const u64 mv = stop ? (on & ((stop & (0ull - stop)) - 1ull)) : on;
            if ((mv >> lane) & 1ull) heap[slot >> 1] = e;
            if (mv == 0) break;
            i = rl(slot, 63 - (int)__builtin_clzll(mv));
            if (stop || __builtin_popcountll(mv) < 5) break;
        }
        if (lane == 0) heap[i] = last;
    };
    int hs = 0;
    const float sum0 = __fadd_rn(__fadd_rn(0.f, t0[0]), t1[0]);
    if (lane == 0) { out_s[0] = sum0; out_k[0] = 0; }       // terms now, keys behind the barrier below
    if (T > 1 && k > 1) {
        push(++hs, __fadd_rn(sum0, __fsub_rn(t0[1], t0[0])), 1);
        push(++hs, __fadd_rn(sum0, __fsub_rn(t1[1], t1[0])), 1 << 16);
    }
    int kk = 1;
    for (; kk < k; kk++) {
        if (hs == 0) break;
        const float s2 = fval(top);
        const int ti = (int)(uint32_t)top;
        const int r0 = ti & 0xffff, r1 = ti >> 16;
        const bool p0 = r0 + 1 < kc && r0 + 1 < T, p1 = r1 + 1 < kc && r1 + 1 < T;
        // the four table entries of this cell's followers: lanes 0-3, requested before the sift-down's round trips
        // (loaded and consumed in every iteration, the stores behind the load: the wait in front of the pushes is then a
        // counted one and the loop head needs none)
        const int tr = (lane & 2) ? r1 : r0;
        const int ti_ = tr + (((lane & 1) && ((lane & 2) ? p1 : p0)) ? 1 : 0);
        const float tv = ((lane & 2) ? t1 : t0)[(lane & 3) == lane ? ti_ : 0];
        if (lane == 0) { out_s[kk] = s2; out_k[kk] = ti; }
        do {
            pop(hs--);
            top = heap[1];
        } while (hs > 0 && (int)(uint32_t)top == ti);
        const int tvb = (int)__float_as_uint(tv);
        const float a0 = __uint_as_float((uint32_t)rl(tvb, 0)), a1 = __uint_as_float((uint32_t)rl(tvb, 1));
        const float b0 = __uint_as_float((uint32_t)rl(tvb, 2)), b1 = __uint_as_float((uint32_t)rl(tvb, 3));
        if (kk < k - 1) {                               // (the last cell's pushes would feed nothing)
            if (p0) push(++hs, __fadd_rn(s2, __fsub_rn(a1, a0)), ti + 1);
            if (p1) push(++hs, __fadd_rn(s2, __fsub_rn(b1, b0)), ti + (1 << 16));
        }
    }
    if (lane == 0)
        for (; kk < k; kk++) { out_s[kk] = 3.402823466e+38f; out_k[kk] = -1; }       // fewer than k cells
    __syncthreads();        // lane 0's terms are visible to the wave: ranks -> sub-quantizer indices, 64 cells at a time
    const int64_t* x0 = si0 + q * T;
    const int64_t* x1 = si1 + q * T;
    for (int j = lane; j < k; j += 64) {
        const int64_t t = out_k[j];
        if (t >= 0) out_k[j] = x0[t & 0xffff] | (x1[t >> 16] << imi_nbits);
    }
}

bool imi_minsum_wide_ok(int T, int k, int kc) {
    return k > 1 && k <= 4096 && T <= 4096 && kc <= 32768;
}

void launch_imi_minsum_wide(const float* sv0, const int64_t* si0, const float* sv1, const int64_t* si1, int T, int64_t nq, int k,
                            int kc, int imi_nbits, float* sums, int64_t* keys, hipStream_t s) {
    if (nq <= 0) return;
    const size_t smem = (size_t)8 * k;
    ensure_dynamic_lds(reinterpret_cast<const void*>(imi_minsum_wide_kernel), smem);
    hipLaunchKernelGGL(imi_minsum_wide_kernel, dim3((unsigned)nq), dim3(64), smem, s, sv0, si0, sv1, si1, T, nq, k, kc, imi_nbits, sums,
                       keys);
}

}  // namespace vlq
