// Shared tail of the scan kernels: merge the NW waves' running selections and
// emit (distance, label) rows.  Scan positions are translated back to
// (probe, offset) with the per-query prefix sums `cum` kept in LDS, then to the
// stored id -- the only place the id array is touched.
#pragma once
#include "kernels.h"
#include "wave_topk.cuh"

namespace vlq {

// resolve(p, lkey, loff): list id and list start offset of probe p
template <int KPL, int NW = 4, int QR = 1, typename Sel, typename Resolve>
__device__ __forceinline__ void merge_and_emit(Sel& sel, unsigned char* smraw,
                                               const uint32_t* cum, const ScanArgs& a, int64_t q,
                                               int wave, int lane, Resolve resolve) {
    sel.flush();
    __syncthreads();                           // LUT buffers are free from here on
    u64* mb = reinterpret_cast<u64*>(smraw);   // [NW][k], aliases the LUT
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = r * 64 + lane;
        if (e < a.k) mb[wave * a.k + e] = sel.best[r];
    }
    __syncthreads();
    if (wave != 0) return;
    for (int w = 1; w < NW; w++)
        for (int e0 = 0; e0 < a.k; e0 += 64) {
            const int e = e0 + lane;
            const bool valid = e < a.k;
            const u64 key = valid ? mb[w * a.k + e] : kMaxKey;
            sel.offer_key(key, valid);
        }
    sel.flush();
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

}  // namespace vlq
