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
#include "wave_topk.cuh"

namespace vlq {

namespace {

constexpr int kPendCap = 1024;      // shared pending queue (keys); a trip appends at most 256

// ascending bitonic sort of a[0..N) in LDS by the 256 threads of the workgroup (N a power of two >= 512)
template <int N>
__device__ __forceinline__ void wg_bitonic_sort(u64* a, int t) {
    for (int size = 2; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int p0 = 0; p0 < N / 2; p0 += 256) {
                const int p = p0 + t;
                const int i = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
                const int j = i | stride;
                const u64 x = a[i], y = a[j];
                const bool up = (i & size) == 0 || size == N;
                if ((x > y) == up) { a[i] = y; a[j] = x; }
            }
            __syncthreads();
        }
    }
}

// a[0..N) bitonic -> ascending
template <int N>
__device__ __forceinline__ void wg_bitonic_merge(u64* a, int t) {
    for (int stride = N >> 1; stride > 0; stride >>= 1) {
#pragma unroll
        for (int p0 = 0; p0 < N / 2; p0 += 256) {
            const int p = p0 + t;
            const int i = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
            const int j = i | stride;
            const u64 x = a[i], y = a[j];
            if (x > y) { a[i] = y; a[j] = x; }
        }
        __syncthreads();
    }
}

}  // namespace

// KC: capacity of the best list (512 or 1024 >= k)
template <int KC, bool IMI>
__global__ __launch_bounds__(256) void scan16_bigk_kernel(ScanArgs a, int lut_region) {
    constexpr int E = 4096, NT = 256, NI = 4, NW = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                         // [2][E] at LDS offsets 0 / 16384
    u64* best = reinterpret_cast<u64*>(smraw + lut_region);               // [KC] the best keys so far (unsorted until the end)
    u64* pend = best + KC;                                                // [kPendCap] queue, contiguous behind them
    ProbeMeta pm;
    pm.carve(reinterpret_cast<unsigned char*>(pend + kPendCap), a.nprobe);
    int32_t* misc = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(pend + kPendCap) +
                                               ProbeMeta::bytes(a.nprobe));    // cut, nlive, npend, thr bits, [2][4] trip counts, select state
    uint16_t* ord = reinterpret_cast<uint16_t*>(misc + 20);                     // [nprobe] visited probes

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
        if (lane == 0) { misc[0] = cut; misc[1] = nl; misc[2] = 0; misc[3] = (int32_t)__float_as_uint(3.402823466e+38f); }
    }
    __syncthreads();
    const int nlive = misc[1];
    float thr = 3.402823466e+38f;          // distance of the k-th best key so far, or FLT_MAX

    // keep the k smallest keys of best[0..KC) + pend[0..n); every thread of the workgroup calls it.
    // misc[12] = digit chosen, misc[13] = rank still wanted inside it, misc[14] = candidates left,
    // misc[15..16] = the k-th key, misc[17] / misc[18] = hole / mover counts
    // Scratch of a flush (digit counts, hole and mover lists: 5 KB) lives in the LUT buffer that is NOT in
    // use: while probe i is scanned from buffer `buf`, the other buffer's table (probe i-1) is consumed and
    // the next one is built only after this probe's last trip.
    int buf = 0;
    auto flush = [&]() {
        unsigned char* scratch = reinterpret_cast<unsigned char*>(lut + (buf ^ 1) * E);
        int32_t* hist = reinterpret_cast<int32_t*>(scratch);                  // [256] radix-select digit counts
        uint16_t* holes = reinterpret_cast<uint16_t*>(hist + 256);            // [KC] slots of the best area that lose their key
        uint16_t* movers = holes + KC;                                        // [kPendCap] queue entries that stay
        const int n = misc[2];
        const int N = KC + n;                      // entries of the contiguous array best | pend
        u64* all = best;
        u64 prefix = 0;                            // digits decided so far (high bytes)
        int want = a.k;                            // 1-based rank of the wanted key among the candidates
        int shift = 56;
        bool unique = false;
        __syncthreads();
        for (; shift >= 0; shift -= 8) {
            hist[t] = 0;
            __syncthreads();
            const u64 himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
            // (the leading bytes of ordered distances are nearly constant: 64 lanes adding to ONE counter would
            // serialise in the LDS atomic unit, so a wave whose candidates all share the digit adds once)
            for (int e0 = 0; e0 < N; e0 += NT) {
                const int e = e0 + t;
                int digit = -1;
                if (e < N) {
                    const u64 key = all[e];
                    if ((key & himask) == prefix) digit = (int)((key >> shift) & 255u);
                }
                const u64 part = __ballot(digit >= 0);
                if (part != 0) {
                    const int src = __builtin_ffsll((long long)part) - 1;
                    const int d0 = __shfl(digit, src, 64);
                    const u64 same = __ballot(digit == d0);
                    if (same == part) { if (lane == src) atomicAdd(&hist[d0], __popcll(part)); }
                    else if (digit >= 0) atomicAdd(&hist[digit], 1);
                }
            }
            __syncthreads();
            if (wave == 0) {                       // the digit whose cumulative count reaches `want`
                const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
                const int own = c0 + c1 + c2 + c3;
                int incl = own;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) {
                    const int o = __shfl_up(incl, sft, 64);
                    if (lane >= sft) incl += o;
                }
                const int before = incl - own;
                if (before < want && want <= incl) {   // exactly one lane
                    int r = want - before, d = 0, cnt = c0;
                    if (r > c0) { r -= c0; d = 1; cnt = c1; if (r > c1) { r -= c1; d = 2; cnt = c2; if (r > c2) { r -= c2; d = 3; cnt = c3; } } }
                    misc[12] = 4 * lane + d;
                    misc[13] = r;
                    misc[14] = cnt;
                }
            }
            __syncthreads();
            prefix |= (u64)(uint32_t)misc[12] << shift;
            want = misc[13];
            if (misc[14] == 1) { unique = true; break; }     // one candidate left: fetch it instead of more passes
        }
        u64 kth = prefix;
        if (unique && shift > 0) {
            const u64 himask = ~0ull << shift;
            for (int e = t; e < N; e += NT) {
                const u64 key = all[e];
                if ((key & himask) == prefix) { misc[15] = (int32_t)(uint32_t)key; misc[16] = (int32_t)(uint32_t)(key >> 32); }
            }
            __syncthreads();
            kth = ((u64)(uint32_t)misc[16] << 32) | (uint32_t)misc[15];
        }
        // keepers: real keys <= kth (exactly k of them unless fewer than k real keys exist: then kth
        // is the padding value and every real key stays).  Queue keepers move into the holes of the
        // best area.
        if (t == 0) { misc[17] = 0; misc[18] = 0; }
        __syncthreads();
        for (int e = t; e < KC; e += NT) {
            const u64 key = all[e];
            if (!(key <= kth && key != kMaxKey)) holes[atomicAdd(&misc[17], 1)] = (uint16_t)e;
        }
        for (int e = t; e < n; e += NT) {
            const u64 key = pend[e];
            if (key <= kth && key != kMaxKey) movers[atomicAdd(&misc[18], 1)] = (uint16_t)e;
        }
        __syncthreads();
        const int nholes = misc[17], nmov = misc[18];
        for (int e = t; e < nholes; e += NT) best[holes[e]] = e < nmov ? pend[movers[e]] : kMaxKey;
        __syncthreads();
        if (t == 0) {
            misc[2] = 0;
            misc[3] = (int32_t)(kth == kMaxKey ? __float_as_uint(3.402823466e+38f)
                                               : __float_as_uint(ordered_to_f32((uint32_t)(kth >> 32))));
        }
        __syncthreads();
        thr = __uint_as_float((uint32_t)misc[3]);
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
    for (int i = 0; i < nlive; i++) {
        const uint32_t len = n_len;
        const float dis0 = n_dis0;
        const uint32_t pos0 = n_pos0;
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + n_off;
        build_lut16<NI>(lut + buf * E, t, t2r, m2t3);
        uint4 cc = c0;
        prefetch(i + 1);
        __syncthreads();
        // trips of 256 consecutive codes; the trip count is the same for every thread
        for (uint32_t j0 = 0; j0 < len; j0 += NT) {
            const uint32_t j = j0 + t;
            const uint4 cn = cp[min(j + NT, len - 1)];
            const float dis = buf == 0 ? adc16_fixed<0>(cc, dis0, two) : adc16_fixed<1>(cc, dis0, two);
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
        buf ^= 1;
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
    const int kc = a.k <= 512 ? 512 : 1024;
    const size_t lutb = (size_t)2 * 4096 * 4;
    const size_t smem = lutb + (size_t)(kc + kPendCap) * 8 + (size_t)a.nprobe * 24 + 8 + 80 + (size_t)a.nprobe * 2 + 64;
    const bool imi = a.imi_nbits > 0;
    if (kc == 512) { if (imi) launch_bigk_t<512, true>(a, (int)lutb, smem, s); else launch_bigk_t<512, false>(a, (int)lutb, smem, s); }
    else { if (imi) launch_bigk_t<1024, true>(a, (int)lutb, smem, s); else launch_bigk_t<1024, false>(a, (int)lutb, smem, s); }
}

}  // namespace vlq
