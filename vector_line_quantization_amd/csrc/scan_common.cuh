// Shared tail of the scan kernels: merge the NW waves' running selections and
// emit (distance, label) rows.  Scan positions are translated back to
// (probe, offset) with the per-query prefix sums `cum` kept in LDS, then to the
// stored id -- the only place the id array is touched.
#pragma once
#include "kernels.h"
#include "wave_topk.cuh"

namespace vlq {

// Rows out: the selection's sorted keys -> (distance, label).  Scan positions are translated with
// the prefix sums `cum`; resolve(p, lkey, loff): list id and list start offset of probe p.
template <int KPL, typename Sel, typename Resolve>
__device__ __forceinline__ void emit_rows(const Sel& sel, const uint32_t* cum, const ScanArgs& a, int64_t q,
                                          int lane, Resolve resolve) {
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = r * 64 + lane;
        if (e >= a.k) continue;
        const u64 key = sel.best[r];
        float dis = 3.402823466e+38f;          // Heap.h:318-321 padding
        int64_t id = -1;
        if (key != kMaxKey) {
            dis = ordered_to_f32((uint32_t)(key >> 32));
            const uint32_t pos = (uint32_t)key;
            int lo = 0, hi = a.nprobe;         // last probe p with cum[p] <= pos
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (cum[mid] <= pos) lo = mid; else hi = mid;
            }
            int64_t lkey, loff;
            resolve(lo, lkey, loff);
            const int64_t o = pos - cum[lo];
            id = a.store_pairs ? (lkey << 32 | o) : a.ids[loff + o];   // IndexIVFPQ.cpp:798
        }
        a.D[q * a.k + e] = dis;
        a.I[q * a.k + e] = id;
    }
}

// joins the NW waves' selections in wave 0 (its `sel` then holds the workgroup's k best keys);
// returns true in wave 0 only
template <int KPL, int NW = 4, int QR = 1, typename Sel>
__device__ __forceinline__ bool merge_waves(Sel& sel, unsigned char* smraw, int k, int wave, int lane) {
    sel.flush();
    __syncthreads();                           // LUT buffers are free from here on
    u64* mb = reinterpret_cast<u64*>(smraw);   // [NW][k], aliases the LUT
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = r * 64 + lane;
        if (e < k) mb[wave * k + e] = sel.best[r];
    }
    __syncthreads();
    if (wave != 0) return false;
    for (int w = 1; w < NW; w++)
        for (int e0 = 0; e0 < k; e0 += 64) {
            const int e = e0 + lane;
            const bool valid = e < k;
            const u64 key = valid ? mb[w * k + e] : kMaxKey;
            sel.offer_key(key, valid);
        }
    sel.flush();
    return true;
}

template <int KPL, int NW = 4, int QR = 1, typename Sel, typename Resolve>
__device__ __forceinline__ void merge_and_emit(Sel& sel, unsigned char* smraw,
                                               const uint32_t* cum, const ScanArgs& a, int64_t q,
                                               int wave, int lane, Resolve resolve) {
    if (!merge_waves<KPL, NW, QR>(sel, smraw, a.k, wave, lane)) return;
    emit_rows<KPL>(sel, cum, a, q, lane, resolve);
}

}  // namespace vlq
