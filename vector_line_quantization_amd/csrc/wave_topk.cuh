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

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int mask) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 umin64(u64 a, u64 b) { return a < b ? a : b; }
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a < b ? b : a; }

// Ascending bitonic sort of 64 keys, one per lane.
__device__ __forceinline__ u64 wave_sort64(u64 key, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            u64 other = shfl_xor_u64(key, stride);
            bool up = (lane & size) == 0;        // size == 64: always ascending
            bool lower = (lane & stride) == 0;
            key = (lower == up) ? umin64(key, other) : umax64(key, other);
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
                u64 a = best[r], b = best[r | rs];
                best[r] = umin64(a, b);
                best[r | rs] = umax64(a, b);
            }
        }
    }
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) {
        bool lower = (lane & stride) == 0;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            u64 other = shfl_xor_u64(best[r], stride);
            best[r] = lower ? umin64(best[r], other) : umax64(best[r], other);
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
                        p[r] = up ? umin64(a, b) : umax64(a, b);
                        p[r | rs] = up ? umax64(a, b) : umin64(a, b);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const u64 other = shfl_xor_u64(p[r], stride);
                    const int e = r * 64 + lane;
                    const bool up = ((e & size) == 0) || size == 64 * R;
                    const bool lower = (lane & stride) == 0;
                    p[r] = (lower == up) ? umin64(p[r], other) : umax64(p[r], other);
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
template <int KPL, int QR = 1>
struct WaveSelect {
    static_assert(QR <= KPL, "the pending queue cannot exceed the best list");
    u64 best[KPL];
    float thr;        // admission threshold (distance of the k-th best, or FLT_MAX)
    int npend;        // entries parked in `queue` (wave-uniform)
    int k;
    int lane;
    u64* queue;

    __device__ __forceinline__ void init(int k_, u64* queue_, int lane_) {
#pragma unroll
        for (int r = 0; r < KPL; r++) best[r] = kMaxKey;
        thr = 3.402823466e+38f;   // FLT_MAX: Heap.h:76-78 neutral element
        npend = 0;
        k = k_;
        lane = lane_;
        queue = queue_;
    }

    // distance part of element k-1 of the sorted best list
    __device__ __forceinline__ void update_threshold() {
        const int kr = (k - 1) >> 6, kl = (k - 1) & 63;
        u64 row = best[0];
#pragma unroll
        for (int r = 1; r < KPL; r++) row = (r == kr) ? best[r] : row;
        u64 kth = shfl_u64(row, kl);
        // a missing k-th (kMaxKey) keeps the threshold at FLT_MAX
        thr = (kth == kMaxKey) ? 3.402823466e+38f : ordered_to_f32((uint32_t)(kth >> 32));
    }

    // merge the parked candidates into the best list
    __device__ __forceinline__ void flush() {
        if (npend == 0) return;
        u64 p[QR];
#pragma unroll
        for (int r = 0; r < QR; r++) p[r] = (r * 64 + lane < npend) ? queue[r * 64 + lane] : kMaxKey;
        wave_sort_multi<QR>(p, lane);
        // pending reversed (element e -> N-1-e) against the tail of the best list: bitonic split
#pragma unroll
        for (int r = 0; r < QR; r++) {
            const u64 rev = shfl_u64(p[QR - 1 - r], 63 - lane);
            best[KPL - QR + r] = umin64(best[KPL - QR + r], rev);
        }
        wave_bitonic_merge<KPL>(best, lane);
        npend = 0;
        update_threshold();
    }

    // one candidate per lane (`valid` lanes only); wave-uniform control flow.
    // ORDERED = true: this wave offers its candidates in increasing position, so a candidate whose
    // distance EQUALS the threshold can never precede the current k-th key and the strict test is
    // exact.  ORDERED = false (positions arrive out of order): equal distances are queued too and
    // the full (distance, position) key decides at the merge; FLT_MAX itself is never admitted
    // (the reference's heap starts at FLT_MAX and admits only `dis < top`, Heap.h:76-78).
    template <bool ORDERED = true>
    __device__ __forceinline__ void offer(float dis, uint32_t pos, bool valid) {
        bool pred = valid && (ORDERED ? dis < thr : (dis <= thr && dis < 3.402823466e+38f));
        u64 mask = __ballot(pred);
        if (mask == 0) return;
        int c = __popcll(mask);
        if (npend + c > 64 * QR) {
            flush();
            pred = pred && (ORDERED ? dis < thr : dis <= thr);
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
