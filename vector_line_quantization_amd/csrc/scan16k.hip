// 16-byte-code list scan for LARGE selections (256 < k <= 1024; the reference's GPU limit is
// k <= 1024, gpu/impl/IVFPQ.cu:966-967).  Same arithmetic as scan16_kernel -- table entries
// term2 + (-2 <q, cent>) (IndexIVFPQ.cpp:641-644), distances dis0 + t[0] + ... + t[15] left to
// right (:788-794), strict admission against the k-th best (Heap.h:68-79) -- but ONE selection
// per workgroup instead of one per wave:
//   * with k of the order of the number of codes a wave sees, four private top-k lists admit
//     ~4 k (1 + ln(n / 4k)) candidates and each of their merges moves 1024 keys through a
//     64-lane network (scan16_kernel<16>: 4.7 ms per 10 000 queries at k = 1000, nearly all of it
//     merges); one shared list admits k (1 + ln(n / k)) and 256 threads merge it;
//   * admitted keys (ordered distance << 32 | scan position: a total order, so the result does
//     not depend on which wave met a code) are appended to a shared LDS queue behind the current
//     best keys; the workgroup walks a list in trips of 256 codes with one barrier per trip, and
//     when the queue may not take another trip the k smallest of (best + queue) are kept: the
//     k-th smallest key is found by an MSB-first radix select (8-bit digits, LDS histogram, stops
//     as soon as one candidate is left), the keepers of the queue move into the holes the losers
//     leave in the best area -- no sort; the best keys are sorted once, at the end;
//   * positions do not arrive in increasing order across waves, so candidates whose distance
//     EQUALS the current k-th distance are queued too and the full key decides (WaveSelect's
//     unordered rule); FLT_MAX itself is never admitted.
#include <type_traits>

#include "kernels.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "walk_order.cuh"
#include "wave_topk.cuh"

namespace vlq {

namespace {

constexpr int kPendCap = 1024;      // shared pending queue (keys); a trip appends at most 256

// wave-wide minimum / inclusive prefix sum without LDS traffic: DPP inside the rows of 16 lanes, readlane across rows
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));     // quad_perm:[1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));     // quad_perm:[2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));    // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));    // row_mirror
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ int wave_inclusive_sum(int x, int lane) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);      // row_shr:1 (lanes without a source keep 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);      // row_shr:8
    const int s0 = __builtin_amdgcn_readlane(x, 15), s1 = __builtin_amdgcn_readlane(x, 31), s2 = __builtin_amdgcn_readlane(x, 47);
    const int row = lane >> 4;
    return x + (row >= 1 ? s0 : 0) + (row >= 2 ? s1 : 0) + (row >= 3 ? s2 : 0);
}

// Ascending bitonic sort of a[0..N) in LDS by the 4 waves of the workgroup (N = 512 or 1024).  A wave keeps its N / 4
// consecutive keys in registers (element r * 64 + lane of its chunk): of the network's stages only the three whose
// partner lies in another wave's chunk go through LDS and a workgroup barrier; strides of 64 and more inside a chunk are
// register pairs, smaller ones lane exchanges.  (One barrier per stage -- 45 / 55 of them -- was 10 % of the k = 1000
// kernel.)
template <int N>
__device__ __forceinline__ void wg_bitonic_sort(u64* a, int t) {
    constexpr int R = N / 256, CH = 64 * R;          // keys per lane, keys per wave
    const int lane = t & 63, gbase = (t >> 6) * CH;
    u64 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = a[gbase + r * 64 + lane];
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= CH) {
#pragma unroll
                for (int r = 0; r < R; r++) a[gbase + r * 64 + lane] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int gi = gbase + r * 64 + lane;
                    const bool up = (gi & size) == 0 || size == N, lower = (gi & stride) == 0;
                    v[r] = pick64(v[r], a[gi ^ stride], lower == up);
                }
                __syncthreads();
            } else if (stride >= 64) {
                const int rs = stride >> 6;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((r & rs) == 0) {
                        const bool up = ((gbase + r * 64) & size) == 0 || size == N;
                        const u64 x = v[r], y = v[r | rs];
                        const bool keep = (x < y) == up;
                        v[r] = keep ? x : y;
                        v[r | rs] = keep ? y : x;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int gi = gbase + r * 64 + lane;
                    const bool up = (gi & size) == 0 || size == N, lower = (lane & stride) == 0;
                    v[r] = pick64(v[r], shfl_xor_u64(v[r], stride), lower == up);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) a[gbase + r * 64 + lane] = v[r];
    __syncthreads();
}

}  // namespace

// KC: capacity of the best list (512 or 1024 >= k)
template <int KC, bool IMI>
__global__ __launch_bounds__(256) void scan16_bigk_kernel(ScanArgs a, int lut_region) {
    constexpr int E = 4096, NT = 256, NI = 4, NW = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                         // [E] at LDS offset 0: ONE table (36.5 KB per workgroup: 4 per CU)
    u64* best = reinterpret_cast<u64*>(smraw + lut_region);               // [KC] the best keys so far (unsorted until the end)
    u64* pend = best + KC;                                                // [kPendCap] queue, contiguous behind them
    int32_t* hist = reinterpret_cast<int32_t*>(pend + kPendCap);          // [3][256] bucket counts of a flush
    u64* small = reinterpret_cast<u64*>(hist + 3 * 256);                  // [64] the last keys of a selection
    ProbeMeta pm;
    pm.carve(reinterpret_cast<unsigned char*>(small + 64), a.nprobe);
    int32_t* misc = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(small + 64) +
                                               ProbeMeta::bytes(a.nprobe));    // cut, nlive, npend, -, [2][4] trip counts, flush state
    uint16_t* ord = reinterpret_cast<uint16_t*>(misc + 26);                     // [nprobe] visited probes (misc[22..25]: min / max key of a flush)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
    int64_t q;
    {
        const int64_t b = blockIdx.x;
        const int64_t s = (b & 7) * a.xcd_chunk + (b >> 3);
        if (s >= a.nq) return;
        q = a.qorder ? a.qorder[s] : s;
    }
    const int64_t* kq = a.keys + q * a.nprobe;
    const bool badkey = probe_meta_fill(a, q, pm, t, NT);
    float4 m2t3[NI];
    load_query_table16<NI>(a, q, t, lane, wave, m2t3);
    for (int e = t; e < KC; e += NT) best[e] = kMaxKey;
    __syncthreads();
    if (wave == 0) {
        const int cut = probe_meta_scan(a, pm, lane);
        __builtin_amdgcn_wave_barrier();
        int nl = 0;
        for (int p0 = 0; p0 < cut; p0 += 64) {
            const int p = p0 + lane;
            const bool lv = p < cut && pm.pkey[p] >= 0;
            const u64 mask = __ballot(lv);
            if (lv) ord[nl + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)p;
            nl += __popcll(mask);
        }
        walk_order_sort(a, pm, ord, nl, lane);
        if (lane == 0) { misc[0] = cut; misc[1] = nl; misc[2] = 0; misc[3] = (int32_t)__float_as_uint(3.402823466e+38f); }
    }
    __syncthreads();
    const int nlive = misc[1];
    float thr = 3.402823466e+38f;          // distance of the k-th best key so far, or FLT_MAX

    // keep the k smallest keys of best[0..KC) + pend[0..n); every thread of the workgroup calls it.
    // The kernel is bound by LDS gathers: while one workgroup flushes, the CU's other workgroups keep the LDS pipe full,
    // and every LDS round trip or barrier of the flush waits behind their requests (measured: ~1000 cycles for a
    // 256-counter prefix scan by one wave).  So the flush is organised as FEW dependent LDS steps (round 3):
    //   * a thread holds its KPT keys of the contiguous array best | pend in registers throughout;
    //   * the k-th smallest key is found by bucket counts over the band the keys actually occupy: lo / hi = smallest
    //     / largest ordered distance (register minima, DPP row reductions), bucket = (key - lo) >> s with s the
    //     smallest shift that maps the band onto 256 buckets, the bucket holding rank k is the next band.  (Byte-wise
    //     radix passes from the top spent two passes on digits nearly every key shares -- 64 lanes adding to two or
    //     three counters.)  One barrier per pass: three counter arrays rotate, EVERY wave reads the counters and finds
    //     the bucket itself (DPP prefix sums, readlane), nothing is broadcast through LDS;
    //   * a band of at most 64 keys (runs of equal distances -- identical codes -- would otherwise cost a pass per 8
    //     bits down to the position bits) is gathered and ranked by every wave with readlane loops;
    //   * keys are unique (the scan position is part of them), so a band of one bucket of width 1 IS the key;
    //   * keepers of the queue are written, as keys, over the front of the queue (every thread has its keys in
    //     registers by then); the j-th hole the losers leave in the best area takes the j-th of them.
    // misc[17] / misc[18] = hole / mover counts, misc[19] = keys gathered, misc[22] / misc[23] = min / max distance.
    int nreal = 0;                             // real keys in the best area (every thread knows it)
    constexpr int KPT = (KC + kPendCap) / NT;  // keys per thread: element i * NT + t of best | pend
    constexpr int KB = KC / NT;                // the first KB of them are best-area slots
    auto flush = [&]() {
        const int n = misc[2];
        const int N = KC + n;                      // entries of the contiguous array best | pend
        if (t == 0) { misc[17] = 0; misc[18] = 0; misc[19] = 0; misc[22] = -1; misc[23] = 0; }
        hist[t] = 0;
        hist[256 + t] = 0;
        u64 kreg[KPT];
#pragma unroll
        for (int i = 0; i < KPT; i++) {
            const int e = i * NT + t;
            kreg[i] = e < N ? best[e] : kMaxKey;
        }
        u64 kth = kMaxKey;                         // fewer than k real keys: the padding value, every real key stays
        __syncthreads();
        if (nreal + n >= a.k) {
            uint32_t mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
            for (int i = 0; i < KPT; i++) {
                const uint32_t hd = (uint32_t)(kreg[i] >> 32);
                if (kreg[i] != kMaxKey) { mn = min(mn, hd); mx = max(mx, hd); }
            }
            mn = wave_min_u32(mn);
            mx = ~wave_min_u32(~mx);
            if (lane == 0) { atomicMin(reinterpret_cast<uint32_t*>(&misc[22]), mn); atomicMax(reinterpret_cast<uint32_t*>(&misc[23]), mx); }
            __syncthreads();
            u64 lo = (u64)(uint32_t)misc[22] << 32, hi = ((u64)(uint32_t)misc[23] << 32) | 0xFFFFFFFFull;
            int want = a.k;                        // 1-based rank of the wanted key among the keys >= lo
            for (int pass = 0;; pass++) {
                const u64 width = hi - lo;
                if (width == 0) { kth = lo; break; }
                const int s = max(0, 56 - (int)__builtin_clzll(width));      // (width >> s) <= 255
                int32_t* H = hist + (pass % 3) * 256;
#pragma unroll
                for (int i = 0; i < KPT; i++) {
                    const u64 key = kreg[i];
                    if (key >= lo && key <= hi) atomicAdd(&H[(int)((key - lo) >> s)], 1);
                }
                __syncthreads();
                // (every wave has left the counters of pass - 1: they become those of pass + 2, needed after the next barrier)
                hist[((pass + 2) % 3) * 256 + t] = 0;
                // the bucket whose cumulative count reaches `want`: lane l owns buckets 4l .. 4l+3
                const int4 c = *reinterpret_cast<const int4*>(H + 4 * lane);
                const int own = c.x + c.y + c.z + c.w;
                const int incl = wave_inclusive_sum(own, lane);
                const int before = incl - own;
                int r = want - before, d = 0, cnt = c.x;
                if (r > c.x) { r -= c.x; d = 1; cnt = c.y; if (r > c.y) { r -= c.y; d = 2; cnt = c.z; if (r > c.z) { r -= c.z; d = 3; cnt = c.w; } } }
                const u64 hit = __ballot(before < want && want <= incl);       // exactly one lane
                const int src = __builtin_ffsll((long long)hit) - 1;
                const int b = __builtin_amdgcn_readlane(4 * lane + d, src);
                want = __builtin_amdgcn_readlane(r, src);
                cnt = __builtin_amdgcn_readlane(cnt, src);
                lo += (u64)(uint32_t)b << s;
                { const u64 top = lo + ((1ull << s) - 1ull); hi = (top >= lo && top < hi) ? top : hi; }     // (top < lo: wrapped)
                if (s == 0) { kth = lo; break; }
                if (cnt <= 64) {
#pragma unroll
                    for (int i = 0; i < KPT; i++)
                        if (kreg[i] >= lo && kreg[i] <= hi) small[atomicAdd(&misc[19], 1)] = kreg[i];
                    __syncthreads();
                    const u64 mine = lane < cnt ? small[lane] : kMaxKey;
                    const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
                    int rank = 0;
                    for (int j = 0; j < cnt; j++) {
                        const u64 o = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)mhi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)mlo, j);
                        rank += o < mine ? 1 : 0;
                    }
                    const u64 m = __ballot(lane < cnt && rank == want - 1);   // exactly one lane
                    const int sl = __builtin_ffsll((long long)m) - 1;
                    kth = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)mhi, sl) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)mlo, sl);
                    break;
                }
            }
        }
        // keepers: real keys <= kth (exactly k of them unless fewer than k real keys exist: then kth is the padding
        // value and every real key stays).  A wave counts its holes and its queue keepers in registers and takes its
        // ranks with one LDS atomic each.
        u64 hb[KB], mb[KPT - KB];
        int nh = 0, nm = 0;
#pragma unroll
        for (int i = 0; i < KB; i++) { hb[i] = __ballot(!(kreg[i] <= kth && kreg[i] != kMaxKey)); nh += __popcll(hb[i]); }
#pragma unroll
        for (int i = KB; i < KPT; i++) { mb[i - KB] = __ballot(kreg[i] <= kth && kreg[i] != kMaxKey); nm += __popcll(mb[i - KB]); }
        int hbase = 0, mbase = 0;
        if (lane == 0) { hbase = atomicAdd(&misc[17], nh); mbase = atomicAdd(&misc[18], nm); }
        hbase = __builtin_amdgcn_readfirstlane(hbase);
        mbase = __builtin_amdgcn_readfirstlane(mbase);
        const u64 below = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = KB; i < KPT; i++) {
            if ((mb[i - KB] >> lane) & 1ull) pend[mbase + __popcll(mb[i - KB] & below)] = kreg[i];
            mbase += __popcll(mb[i - KB]);
        }
        __syncthreads();
        const int nholes = misc[17], nmov = misc[18];
#pragma unroll
        for (int i = 0; i < KB; i++) {
            if ((hb[i] >> lane) & 1ull) {
                const int j = hbase + __popcll(hb[i] & below);
                best[i * NT + t] = j < nmov ? pend[j] : kMaxKey;
            }
            hbase += __popcll(hb[i]);
        }
        nreal = KC - nholes + min(nholes, nmov);
        thr = kth == kMaxKey ? 3.402823466e+38f : ordered_to_f32((uint32_t)(kth >> 32));
        if (t == 0) misc[2] = 0;
        __syncthreads();
    };

    float4 t2r[NI];
    uint4 c0 = make_uint4(0, 0, 0, 0);
    uint32_t n_len = 0, n_pos0 = 0;
    float n_dis0 = 0.f;
    int64_t n_off = 0;
    auto prefetch = [&](int i) {
        if (i >= nlive) return;
        const int p = ord[i];
        const int64_t key = pm.pkey[p];
        n_len = __builtin_amdgcn_readfirstlane(pm.plen[p]);
        n_dis0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pm.pd0[p])));
        n_pos0 = __builtin_amdgcn_readfirstlane(pm.cum[p]);
        {
            const int64_t o = pm.poff[p];
            n_off = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)o >> 32)) << 32) |
                              __builtin_amdgcn_readfirstlane((uint32_t)o));
        }
        if (IMI) {
            const int64_t ki0 = key & ((int64_t(1) << a.imi_nbits) - 1), ki1 = key >> a.imi_nbits;
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) {
                const int64_t ki = (NW * i2 + wave) < 8 ? ki0 : ki1;
                t2r[i2] = reinterpret_cast<const float4*>(a.term2 + (size_t)ki * E)[i2 * NT + t];
            }
        } else {
            const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)key * E);
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) t2r[i2] = src[i2 * NT + t];
        }
        c0 = (reinterpret_cast<const uint4*>(a.codes) + n_off)[min((uint32_t)t, n_len - 1)];
    };
    prefetch(0);
    uint64_t nscan = 0;
    // Queue fill as every thread knows it.  The slot of a key comes from an LDS atomic on misc[2], but the
    // decision to flush must be the same in all four waves, and a wave that is already in the next trip
    // may have bumped misc[2] again: so each wave also posts its count of the trip in a slot of the
    // trip's parity, and after the trip's barrier everyone adds the four counts of THAT parity.
    int npend_reg = 0;
    uint32_t trip = 0;
    bool tripped = false;
    for (int i = 0; i < nlive; i++) {
        const uint32_t len = n_len;
        const float dis0 = n_dis0;
        const uint32_t pos0 = n_pos0;
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + n_off;
        if (i > 0 && !tripped) __syncthreads();                // (an empty list before: no trip barrier has said that everyone is done with the table)
        __builtin_amdgcn_s_setprio(2);                         // as in scan16.hip: the barrier below waits for the slowest builder
        build_lut16<NI>(lut, t, t2r, m2t3);
        uint4 cc = c0;
        prefetch(i + 1);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        // trips of 256 consecutive codes; the trip count is the same for every thread
        for (uint32_t j0 = 0; j0 < len; j0 += NT) {
            const uint32_t j = j0 + t;
            const uint4 cn = cp[min(j + NT, len - 1)];
            const float dis = adc16_halves<0>(cc, dis0, two);
            const bool pred = j < len && dis <= thr && dis < 3.402823466e+38f;
            const u64 mask = __ballot(pred);
            const int cnt = __popcll(mask);
            int32_t* wcnt = misc + 4 + 4 * (trip & 1u);
            if (lane == 0) wcnt[wave] = cnt;
            if (mask != 0) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&misc[2], cnt);
                base = __builtin_amdgcn_readfirstlane(base);
                if (pred) pend[base + __popcll(mask & ((1ull << lane) - 1ull))] = make_key(dis, pos0 + j);
            }
            cc = cn;
            __syncthreads();                                  // this trip's appends are in
            npend_reg += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            trip++;
            if (npend_reg > kPendCap - NT) { flush(); npend_reg = 0; }   // the same decision in every thread
        }
        nscan += len;
        tripped = len > 0;
    }
    flush();
    wg_bitonic_sort<KC>(best, t);                             // the one sort: rows leave in ascending key order
    // rows out
    for (int e = t; e < a.k; e += NT) {
        const u64 key = best[e];
        float dis = 3.402823466e+38f;          // Heap.h:318-321 padding
        int64_t id = -1;
        if (key != kMaxKey) {
            dis = ordered_to_f32((uint32_t)(key >> 32));
            const uint32_t pos = (uint32_t)key;
            int lo = 0, hi = a.nprobe;         // last probe p with cum[p] <= pos
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (pm.cum[mid] <= pos) lo = mid; else hi = mid;
            }
            const int64_t o = pos - pm.cum[lo];
            id = a.store_pairs ? (kq[lo] << 32 | o) : a.ids[pm.poff[lo] + o];   // IndexIVFPQ.cpp:798
        }
        a.D[q * a.k + e] = dis;
        a.I[q * a.k + e] = id;
    }
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
    if (badkey) *a.bad_key = 1;
}

template <int KC, bool IMI>
static void launch_bigk_t(const ScanArgs& a, int lut_region, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan16_bigk_kernel<KC, IMI>), smem);
    hipLaunchKernelGGL((scan16_bigk_kernel<KC, IMI>), dim3((unsigned)(8 * a.xcd_chunk)), dim3(256), smem, s, a, lut_region);
}

void launch_scan16_bigk(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    a.nsplit = 1;
    a.xcd_chunk = (int)((a.nq + 7) / 8);
    const int kc = a.k <= 256 ? 256 : a.k <= 512 ? 512 : 1024;
    const size_t lutb = (size_t)4096 * 4;
    const size_t smem = lutb + (size_t)(kc + kPendCap) * 8 + 3 * 256 * 4 + 64 * 8 + (size_t)a.nprobe * 24 + 8 + 104 + (size_t)a.nprobe * 2 + 64;
    const bool imi = a.imi_nbits > 0;
    if (kc == 256) { if (imi) launch_bigk_t<256, true>(a, (int)lutb, smem, s); else launch_bigk_t<256, false>(a, (int)lutb, smem, s); }
    else if (kc == 512) { if (imi) launch_bigk_t<512, true>(a, (int)lutb, smem, s); else launch_bigk_t<512, false>(a, (int)lutb, smem, s); }
    else { if (imi) launch_bigk_t<1024, true>(a, (int)lutb, smem, s); else launch_bigk_t<1024, false>(a, (int)lutb, smem, s); }
}

}  // namespace vlq
