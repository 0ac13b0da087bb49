// Wave64-native running k-selection (the k smallest of a stream), CDNA4.
//
// Replaces the role of the reference's WarpSelect/BlockSelect (gpu/utils/Select.cuh,
// hard-wired to 32 lanes, gpu/utils/DeviceDefs.cuh:15-26) with a design written for
// 64-lane wavefronts:
//   * every candidate is one 64-bit key  (order-preserving image of the f32
//     distance) << 32 | scan position, so the selection is a TOTAL order
//     (distance, position) and the result does not depend on how the stream was
//     split over lanes, waves or workgroups;
//   * a wave keeps its current best N = 64*KPL keys sorted across lanes
//     (element e = r*64 + lane lives in register r of lane `lane`), the k-th of
//     them is the admission threshold;
//   * admitted candidates are parked in a 64-entry LDS queue and merged in bulk
//     (bitonic sort of the queue + one bitonic merge with the best list), so the
//     common iteration costs one compare and one ballot.
// Admission is `dis < threshold` -- the strict test of the reference's heap
// (IndexIVFPQ.cpp:796, Heap.h:68-79): later equal distances never displace
// earlier ones.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlq {

typedef unsigned long long u64;

static constexpr u64 kMaxKey = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ uint32_t f32_to_ordered(float f) {
    uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(uint32_t u) {
    uint32_t b = u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(b);
}
__device__ __forceinline__ u64 make_key(float dis, uint32_t pos) {
    return ((u64)f32_to_ordered(dis) << 32) | pos;
}

// value of lane (lane ^ stride); stride is a compile-time constant after unrolling.  Every stride is a VALU
// move -- no trip through the LDS crossbar, whose queue is where the list scans' gathers wait: a ds_swizzle or
// ds_bpermute issued by a selection under a saturated LDS pipe returns after hundreds of cycles.  1, 2: quad_perm;
// 4: two bank-masked row shifts; 8: row_ror:8; 16 / 32: gfx950's v_permlane16_swap / v_permlane32_swap (with
// both operands = v the swap leaves lane^16 resp. lane^32 in one of the two results, by lane half).
__device__ __forceinline__ uint32_t wave_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v, int stride) {
    switch (stride) {
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
        case 8: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);   // row_ror:8
        case 4: {   // lanes 0-3, 8-11 of a row take lane+4 (row_shl:4, banks 0 and 2), the others lane-4 (row_shr:4, banks 1 and 3)
            const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);
            return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);
        }
        case 16: {  // rows 1, 3 of r[0] <-> rows 0, 2 of r[1]
            const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
            return (wave_lane_id() & 16u) ? r[0] : r[1];
        }
        default: {  // 32: upper half of r[0] <-> lower half of r[1]
            const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
            return (wave_lane_id() & 32u) ? r[0] : r[1];
        }
    }
}
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int mask) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = lane_xor_u32(lo, mask);
    hi = lane_xor_u32(hi, mask);
    return ((u64)hi << 32) | lo;
}
// value of lane 63 - lane: row_mirror inside the rows of 16, then the rows exchanged
__device__ __forceinline__ u64 lane_reverse_u64(u64 v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x140, 0xF, 0xF, true);
    hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x140, 0xF, 0xF, true);
    lo = lane_xor_u32(lane_xor_u32(lo, 16), 32);
    hi = lane_xor_u32(lane_xor_u32(hi, 16), 32);
    return ((u64)hi << 32) | lo;
}
// value of lane `src`, src wave-uniform (v_readlane)
__device__ __forceinline__ u64 bcast_u64(u64 v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}
// compare-exchange results with one 64-bit compare: the smaller (want_min) or larger of (a, b)
__device__ __forceinline__ u64 pick64(u64 a, u64 b, bool want_min) { return ((a < b) == want_min) ? a : b; }
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 umin64(u64 a, u64 b) { return a < b ? a : b; }
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a < b ? b : a; }

// Ascending bitonic sort of 64 32-bit keys, one per lane (VALU only, like the 64-bit one).
__device__ __forceinline__ uint32_t wave_sort64_u32(uint32_t key, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const uint32_t other = lane_xor_u32(key, stride);
            const bool up = (lane & size) == 0;
            const bool lower = (lane & stride) == 0;
            key = (lower == up) ? min(key, other) : max(key, other);
        }
    }
    return key;
}

// Inclusive prefix sum over the 64 lanes, VALU only (DPP row shifts, then the row totals by row_bcast:15 / :31).
__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3
    return v;
}

// Ascending bitonic sort of 64 keys, one per lane.
__device__ __forceinline__ u64 wave_sort64(u64 key, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            u64 other = shfl_xor_u64(key, stride);
            bool up = (lane & size) == 0;        // size == 64: always ascending
            bool lower = (lane & stride) == 0;
            key = pick64(key, other, lower == up);
        }
    }
    return key;
}

// Bitonic merge of a bitonic sequence of N = 64*KPL keys (element e = r*64+lane)
// into ascending order.
template <int KPL>
__device__ __forceinline__ void wave_bitonic_merge(u64 (&best)[KPL], int lane) {
#pragma unroll
    for (int rs = KPL >> 1; rs > 0; rs >>= 1) {   // strides >= 64: register pairs
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            if ((r & rs) == 0) {
                const u64 a = best[r], b = best[r | rs];
                const bool lt = a < b;
                best[r] = lt ? a : b;
                best[r | rs] = lt ? b : a;
            }
        }
    }
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) {
        bool lower = (lane & stride) == 0;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            u64 other = shfl_xor_u64(best[r], stride);
            best[r] = pick64(best[r], other, lower);
        }
    }
}

// Ascending bitonic sort of N = 64*R keys (element e = r*64 + lane).
template <int R>
__device__ __forceinline__ void wave_sort_multi(u64 (&p)[R], int lane) {
    if (R == 1) { p[0] = wave_sort64(p[0], lane); return; }
#pragma unroll
    for (int size = 2; size <= 64 * R; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {                 // partner in another register, same lane
                const int rs = stride >> 6;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((r & rs) == 0) {
                        const bool up = (((r * 64) & size) == 0) || size == 64 * R;
                        const u64 a = p[r], b = p[r | rs];
                        const bool lt = (a < b) == up;      // up is a compile-time constant here
                        p[r] = lt ? a : b;
                        p[r | rs] = lt ? b : a;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const u64 other = shfl_xor_u64(p[r], stride);
                    const int e = r * 64 + lane;
                    const bool up = ((e & size) == 0) || size == 64 * R;
                    const bool lower = (lane & stride) == 0;
                    p[r] = pick64(p[r], other, lower == up);
                }
            }
        }
    }
}

// Running selection state of one wave.  `queue` = 64 u64 slots of LDS owned by
// this wave.
// QR = pending-queue capacity in units of 64 keys (LDS: 64*QR u64 per wave).  A flush costs one
// sort of the pending keys plus one merge of the whole best list, so long lists (KPL >= 8) want
// QR = 4: four times fewer merges of 512..1024 keys.
// OUTLINE: the merge runs as a real function call (values in, values out -- all in registers), so
// the register allocation of the caller's hot loop does not see the ~40 temporaries of the sort /
// merge network; whatever has to be saved around the call is saved in the rare branch only.
template <int KPL>
struct BestList { u64 v[KPL]; float kth; uint32_t kth_pos; };

// (the parked keys travel by value too: read through the callee's generic pointer they were FLAT
// loads whose wait drained the caller's outstanding global prefetches at every flush)
template <int QR>
struct Pending { u64 v[QR]; };

template <int KPL, int QR>
__device__ __forceinline__ BestList<KPL> flush_body(BestList<KPL> b, Pending<QR> pend, int k, int lane) {
    u64 (&p)[QR] = pend.v;
    wave_sort_multi<QR>(p, lane);
    // pending reversed (element e -> N-1-e) against the tail of the best list: bitonic split
#pragma unroll
    for (int r = 0; r < QR; r++) {
        const u64 rev = lane_reverse_u64(p[QR - 1 - r]);
        b.v[KPL - QR + r] = umin64(b.v[KPL - QR + r], rev);
    }
    wave_bitonic_merge<KPL>(b.v, lane);
    // distance part of element k-1 of the sorted best list
    const int kr = (k - 1) >> 6, kl = (k - 1) & 63;
    u64 row = b.v[0];
#pragma unroll
    for (int r = 1; r < KPL; r++) row = (r == kr) ? b.v[r] : row;
    const u64 kth = bcast_u64(row, kl);
    // a missing k-th (kMaxKey) keeps the threshold at FLT_MAX
    b.kth = (kth == kMaxKey) ? 3.402823466e+38f : ordered_to_f32((uint32_t)(kth >> 32));
    b.kth_pos = (kth == kMaxKey) ? 0u : (uint32_t)kth;
    return b;
}
template <int KPL, int QR>
__device__ __noinline__ BestList<KPL> flush_call(BestList<KPL> b, Pending<QR> pend, int k, int lane) {
    return flush_body<KPL, QR>(b, pend, k, lane);
}

template <int KPL, int QR = 1, bool OUTLINE = false>
struct WaveSelect {
    static_assert(QR <= KPL, "the pending queue cannot exceed the best list");
    u64 best[KPL];
    float thr;        // admission threshold: min(thr_own, bound shared by the workgroup)
    float thr_own;    // distance of this wave's k-th best, or FLT_MAX
    uint32_t pos_own; // ... and its position (0 with FLT_MAX: nothing precedes it)
    float thr_le;     // keyed admission (offer_keyed): nothing above min(thr_own, shared minimum) can be admitted
    int npend;        // entries parked in `queue` (wave-uniform)
    int k;
    int lane;
    u64* queue;
    // Optional bound shared by the waves of a workgroup that select from disjoint parts of ONE stream
    // (their lists are merged afterwards): the minimum of their k-th distances.  A wave whose k-th
    // best is T holds k keys with distance <= T, so a candidate of ANOTHER wave can still reach the
    // merged top-k only if dis <= T.  `dis <= T` is `dis < nextup(T)`, so the hot loop keeps its one
    // strict compare; the own threshold stays strict (min(thr_own, nextup(T)) == thr_own when
    // T == thr_own).  The selection itself never touches the shared word (through a generic pointer
    // that is a FLAT access whose wait also drains the caller's outstanding global loads): the owner
    // of the word -- the kernel, with plain LDS instructions -- publishes `thr_own` when `dirty` and
    // hands the current minimum to refresh_with().
    float thr_sh;     // nextup of the last shared minimum seen (+inf: none)
    float t_sh;       // the last shared minimum itself
    bool dirty;       // thr_own changed since the caller last published it

    __device__ __forceinline__ void init(int k_, u64* queue_, int lane_) {
#pragma unroll
        for (int r = 0; r < KPL; r++) best[r] = kMaxKey;
        thr = thr_own = thr_le = 3.402823466e+38f;   // FLT_MAX: Heap.h:76-78 neutral element
        pos_own = 0;
        t_sh = __builtin_inff();
        thr_sh = __builtin_inff();
        dirty = false;
        npend = 0;
        k = k_;
        lane = lane_;
        queue = queue_;
    }

    // merge the parked candidates into the best list
    __device__ __forceinline__ void flush() {
        if (npend == 0) return;
        BestList<KPL> b;
#pragma unroll
        for (int r = 0; r < KPL; r++) b.v[r] = best[r];
        Pending<QR> pend;
#pragma unroll
        for (int r = 0; r < QR; r++) pend.v[r] = (r * 64 + lane < npend) ? queue[r * 64 + lane] : kMaxKey;
        b = OUTLINE ? flush_call<KPL, QR>(b, pend, k, lane) : flush_body<KPL, QR>(b, pend, k, lane);
#pragma unroll
        for (int r = 0; r < KPL; r++) best[r] = b.v[r];
        npend = 0;
        // (wave-uniform: keep it in a scalar register)
        thr_own = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(b.kth)));
        pos_own = __builtin_amdgcn_readfirstlane(b.kth_pos);
        thr = fminf(thr_own, thr_sh);
        thr_le = fminf(thr_own, t_sh);
        dirty = true;
    }

    // the workgroup's current minimum (ordered image), read by the caller
    __device__ __forceinline__ void refresh_with(uint32_t shared_ordered) {
        const float t = ordered_to_f32(shared_ordered) + 0.0f;      // -0 -> +0: nextup(-0) must be > +0
        thr_sh = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ordered_to_f32(f32_to_ordered(t) + 1u))));
        t_sh = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(t)));
        thr = fminf(thr_own, thr_sh);
        thr_le = fminf(thr_own, t_sh);
    }

    // one candidate per lane (`valid` lanes only); wave-uniform control flow.
    // ORDERED = true: this wave offers its candidates in increasing position, so a candidate whose
    // distance EQUALS the threshold can never precede the current k-th key and the strict test is
    // exact.  ORDERED = false (positions arrive out of order): equal distances are queued too and
    // the full (distance, position) key decides at the merge; FLT_MAX itself is never admitted
    // (the reference's heap starts at FLT_MAX and admits only `dis < top`, Heap.h:76-78).
    template <bool ORDERED = true, bool KEYED = false>
    __device__ __forceinline__ void offer(float dis, uint32_t pos, bool valid) {
        auto admit = [&](bool first) __attribute__((always_inline)) {
            if (KEYED) return dis < thr || (dis == thr_own && pos < pos_own);
            return ORDERED ? dis < thr : (dis <= thr && (!first || dis < 3.402823466e+38f));
        };
        // (keyed: one compare in the scan loop -- whatever is admitted lies at or under min(own k-th, shared minimum); the
        // position is looked at only when a lane passes)
        if (KEYED && __ballot(valid && dis <= thr_le) == 0) return;
        bool pred = valid && admit(true);
        u64 mask = __ballot(pred);
        if (mask == 0) return;
        int c = __popcll(mask);
        if (npend + c > 64 * QR) {
            flush();
            pred = pred && admit(false);
            mask = __ballot(pred);
            if (mask == 0) return;
            c = __popcll(mask);
        }
        if (pred) {
            int slot = npend + __popcll(mask & ((1ull << lane) - 1ull));
            queue[slot] = make_key(dis, pos);
        }
        npend += c;
    }
    // The list scans: a wave's candidates arrive in ANY order of their positions (the walking order of the lists, a long
    // list's further chunks before its first ones), so a candidate AT the wave's k-th distance is admitted when its position
    // precedes the k-th key's -- the full (distance, position) order, the one the reference's heap leaves behind
    // (`dis < top` in scan order: among equal distances the first scanned stays, Heap.h:76-78).  Until round 5 the scans
    // used the ordered rule: a tie at the k-th distance scanned earlier by the reference but visited later here was dropped
    // (tests/test_gpu_ties.py).  FLT_MAX is never admitted: thr_own = FLT_MAX comes with pos_own = 0.
#ifdef VLQ_KEYED_OFF      // (timing A/B only: the ordered rule drops ties, tests/test_gpu_ties.py)
    __device__ __forceinline__ void offer_keyed(float dis, uint32_t pos, bool valid) { offer<true, false>(dis, pos, valid); }
#else
    __device__ __forceinline__ void offer_keyed(float dis, uint32_t pos, bool valid) { offer<false, true>(dis, pos, valid); }
#endif

    // same, for ready-made keys (block-level merge of per-wave results)
    __device__ __forceinline__ void offer_key(u64 key, bool valid) {
        bool pred = valid && key != kMaxKey;
        u64 mask = __ballot(pred);
        if (mask == 0) return;
        int c = __popcll(mask);
        if (npend + c > 64 * QR) flush();
        if (pred) {
            int slot = npend + __popcll(mask & ((1ull << lane) - 1ull));
            queue[slot] = key;
        }
        npend += c;
    }
};

}  // namespace vlq
