// Walking order of a query's probes in the scan kernels (round 4).
//
// The reference scans a query's lists in coarse-distance order (IndexIVFPQ.cpp:983-1060); what a result depends on is each
// code's distance and its scan POSITION (the tie order of the heap replay), and positions are fixed per probe before the walk
// starts (ProbeMeta::cum).  The order in which a workgroup actually visits its probes is therefore free -- the list-owned
// schedule already uses that.  Two steps, both measured on the headline data (10 000 queries, nprobe 32, k 10; the kernel
// was bound by the 16 KB table rows crossing the fabric into the XCDs: 5.38 GB per launch, L2 hit 36 %):
//  1. the nearest `first` probes stay in front (they tighten the admission bound early), the rest is visited in ascending
//     list id.  Queries that run next to each other on an XCD probe overlapping sets of lists; when every workgroup walks its
//     set in the same global order, two workgroups that started at about the same time ask for a shared row at about the
//     same time and the second request finds it in the XCD's L2.  Scan 0.725 -> 0.638 ms, 4.37 GB, L2 hit 48 %
//     (all probes by id 0.641; 2 / 4 / 8 in front 0.636 / 0.647 / 0.668);
//  2. the walk is cyclic and starts at the id a global clock (s_memrealtime, 100 MHz, one counter for the chip) points at,
//     period = the time a walk takes, measured by the workgroups themselves (walk_state): whenever a workgroup starts, it
//     joins the others of its XCD near the same id.  0.638 -> 0.597 ms, 3.43 GB, L2 hit 59 %; nprobe 64 1.24 -> 1.08 ms,
//     nprobe 128 2.52 -> 2.04 ms (coarse-distance order: 1.47 / 3.02).  Period = 0.6 / 0.8 x the measured walk time: 0.619 /
//     0.600; 1.2 / 1.5 x: 0.603 / 0.606.  Every workgroup of a launch must use the SAME period (the clock counts from boot:
//     a period that differs by 0.2 % is a random phase -- a running mean read live measured 0.75 ms).
// What did not work: publishing the position through device-scope atomics (one store per probe to a line per XCD: 1.69 ms);
// non-temporal code loads to keep the rows in L2 longer (0.72: neighbours share the codes too); centring a workgroup's
// walk on the clock by its share of the codes (no change); 8-byte codes / float16 rows (8 KB rows: 0.371 -> 0.387) and
// 64-byte codes (one workgroup per CU: 3.84 -> 3.89) -- the rule in api.hip keeps those in coarse-distance order.
#pragma once
#include "scan16_common.cuh"

namespace vlq {

// What the walking order reads from global memory, requested at the top of the kernel (round 5): three dependent-free loads
// whose latency used to sit between the two set-up barriers (profiles/r05_scan16_phases.txt).
struct WalkPre {
    int flag = 0;       // lane < 32: walk_stat_kernel's count of workgroup `lane`
    int mean_now = 0;   // running mean of the walk time, as read now
    int dur = 0;        // its value when this launch was prepared: the clock period of every workgroup of the launch
};
__device__ __forceinline__ WalkPre walk_prefetch(const ScanArgs& a, int lane) {
    WalkPre w;
    if (a.walk_first < 0) return w;
    if (a.walk_flag && lane < 32) w.flag = a.walk_flag[lane];
    if (a.walk_state) {
        w.mean_now = a.walk_state[(blockIdx.x & 7) * 16];
        // plain accesses: a stale value is as good as a fresh one, and device-scope atomics leave the XCD (measured: +17 %)
        w.dur = a.walk_state[(blockIdx.x & 7) * 16 + 1];       // the same value for every workgroup of this launch
    }
    return w;
}

// wave 0 of the workgroup, after ord[0 .. nl) holds the live probes in coarse-distance order
// returns -1 when the coarse-distance order stays, else the running mean of the walk time as read now (0 = none yet): the
// value walk_state_finish folds this workgroup's own time into
__device__ __forceinline__ int walk_order_sort(const ScanArgs& a, const ProbeMeta& pm, uint16_t* ord, int nl, int lane, const WalkPre& pre) {
    const int first = a.walk_first;
    if (first < 0 || nl - first < 2 || nl - first > 256) return -1;
    if (a.walk_flag) {      // walk_stat_kernel: do this batch's neighbours share most of their lists anyway?
        int v = pre.flag;
        v += (int)lane_xor_u32((uint32_t)v, 16);      // (VALU moves: wave_topk.cuh)
        v += (int)lane_xor_u32((uint32_t)v, 8);
        v += (int)lane_xor_u32((uint32_t)v, 4);
        v += (int)lane_xor_u32((uint32_t)v, 2);
        v += (int)lane_xor_u32((uint32_t)v, 1);
        if (__builtin_amdgcn_readfirstlane(v) > a.walk_limit) return -1;
    }
    __builtin_amdgcn_wave_barrier();
    const int n = nl - first;
    // the clock's share of the list ids (both forms below)
    long long period = a.walk_clock;
    const int mean_now = a.walk_state ? pre.mean_now : 0;
    if (period == 0 && a.walk_state) period = (long long)pre.dur * a.walk_scale / 1000;
    int X = -1;
    if (period > 0 && period < (1ll << 31)) {
        // the low 32 bits of the 100 MHz clock (they wrap every 43 s: one odd phase, speed only), 32-bit remainder, the
        // share of the list ids in float: the 64-bit remainder and quotient by a variable cost two software divisions
        // between the set-up barriers
        const uint32_t r = (uint32_t)wall_clock64() % (uint32_t)period;
        X = (int)(((float)r / (float)(uint32_t)period) * (float)a.nlist);
    }
    if (n <= 64 && a.nlist <= (1 << 22)) {
        // one probe per lane (round 5): a 64-lane sort of (list id, probe) instead of n rounds of v_readlane + compares
        // (12 600 cycles of every workgroup's set-up on the headline shape, wave 1 waiting at the barrier)
        const int pr = lane < n ? ord[first + lane] : 0;
        const uint32_t k32 = lane < n ? (((uint32_t)pm.pkey[pr] << 10) | (uint32_t)pr) : 0xffffffffu;
        const uint32_t sk = wave_sort64_u32(k32, lane);
        const int rot = X >= 0 ? __popcll(__ballot(lane < n && (int)(sk >> 10) < X)) : 0;
        __builtin_amdgcn_wave_barrier();
        if (lane < n) { int pos = lane - rot; if (pos < 0) pos += n; ord[first + pos] = (uint16_t)(sk & 1023u); }
        __builtin_amdgcn_wave_barrier();
        return max(mean_now, 0);
    }
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
    // The cyclic walk starts at the id a global clock points at: period = the walk time the XCD's workgroups measured for
    // themselves in the launches before (walk_state: slot 0 running mean, slot 1 = its value when this launch was prepared --
    // every workgroup of a launch must divide the clock by the SAME period, the clock counts from boot), so that at any moment
    // the workgroups of an XCD, whenever they started, are near the same id and a row shared by two of them is asked for twice
    // within L2's memory.
    int rot = 0;
    if (X >= 0) {
#pragma unroll
        for (int c = 0; c < 4; c++) rot += __popcll(__ballot(lane + 64 * c < n && key[c] < X));
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (lane + 64 * c < n) { int pos = rank[c] - rot; if (pos < 0) pos += n; ord[first + pos] = (uint16_t)p[c]; }
    __builtin_amdgcn_wave_barrier();
    return max(mean_now, 0);
}
__device__ __forceinline__ int walk_order_sort(const ScanArgs& a, const ProbeMeta& pm, uint16_t* ord, int nl, int lane) {
    return walk_order_sort(a, pm, ord, nl, lane, walk_prefetch(a, lane));
}

// thread 0 of a workgroup after its walk: running mean (1/8) of the walk time in clock ticks, per XCD.  `mean` was read at the
// start of the workgroup (a load here would sit between the last probe and the release of the workgroup's slot: measured
// +2 % on the scan); the store is not waited for.
__device__ __forceinline__ void walk_state_finish(const ScanArgs& a, unsigned long long t_begin, int nprobes_walked, int mean) {
    if (!a.walk_state || mean < 0 || nprobes_walked < 8) return;
    const long long d = (long long)(wall_clock64() - t_begin);
    if (d < 100 || d > 100000000) return;
    a.walk_state[(blockIdx.x & 7) * 16] = mean > 0 ? mean + (int)((d - mean) / 8) : (int)d;
}

}  // namespace vlq
