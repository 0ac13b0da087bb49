// Walking order of a query's probes in the scan kernels (round 4).
//
// The reference scans a query's lists in coarse-distance order (IndexIVFPQ.cpp:983-1060); what a result depends on is each
// code's distance and its scan POSITION (the tie order of the heap replay), and positions are fixed per probe before the walk
// starts (ProbeMeta::cum).  The order in which a workgroup actually visits its probes is therefore free -- the list-owned
// schedule already uses that.  Here: the nearest `first` probes stay in front (they tighten the admission bound early), the
// rest is visited in ascending list id.  Queries that run next to each other on an XCD probe overlapping sets of lists; when
// every workgroup walks its set in the same global order, two workgroups that started at about the same time ask for a shared
// 16 KB table row at about the same time, and the second request finds it in the XCD's L2 instead of crossing the fabric.
// Headline data (10 000 queries, nprobe 32): scan 0.725 -> 0.632 ms; all probes by id (first = 0) 0.641; 2 / 4 / 8 in front
// 0.636 / 0.647 / 0.668.  (A walk that STARTS where the XCD's other workgroups currently are -- all workgroups on the same
// rows at once -- was measured too: 1.69 ms.)
#pragma once
#include "scan16_common.cuh"

namespace vlq {

// wave 0 of the workgroup, after ord[0 .. nl) holds the live probes in coarse-distance order
__device__ __forceinline__ void walk_order_sort(const ScanArgs& a, const ProbeMeta& pm, uint16_t* ord, int nl, int lane) {
    const int first = a.walk_first;
    if (first < 0 || nl - first < 2 || nl - first > 256) return;
    if (a.walk_flag) {      // walk_stat_kernel: do this batch's neighbours share most of their lists anyway?
        int v = lane < 32 ? a.walk_flag[lane] : 0;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (__builtin_amdgcn_readfirstlane(v) > a.walk_limit) return;
    }
    __builtin_amdgcn_wave_barrier();
    const int n = nl - first;
    int p[4], key[4], rank[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int idx = lane + 64 * c;
        p[c] = idx < n ? ord[first + idx] : 0;
        key[c] = idx < n ? pm.pkey[p[c]] : 0x7fffffff;
        rank[c] = 0;
    }
#pragma unroll
    for (int c2 = 0; c2 < 4; c2++) {
        if (64 * c2 >= n) break;
        const int jn = min(64, n - 64 * c2);
        for (int j = 0; j < jn; j++) {
            const int kj = __builtin_amdgcn_readlane(key[c2], j);
#pragma unroll
            for (int c = 0; c < 4; c++)
                rank[c] += (kj < key[c] || (kj == key[c] && 64 * c2 + j < 64 * c + lane)) ? 1 : 0;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (lane + 64 * c < n) ord[first + rank[c]] = (uint16_t)p[c];
    __builtin_amdgcn_wave_barrier();
}

}  // namespace vlq
