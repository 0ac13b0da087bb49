// Paired-query variant of the 16-byte-code scan (M = 16, ksub = 256, table mode 1).
//
// The scan of BASELINE config 1 is bound by bytes, not arithmetic: every (query, probe)
// pair needs the 16 KB term2[list] row (5.2 GB per 10 000-query batch, against 1.3-3.6 GB
// of codes), and a CU can pull only so many bytes per clock from L2 / Infinity Cache.
// Queries that are adjacent in the nearest-centroid order probe mostly the SAME lists, so
// one workgroup serves TWO such queries: it walks the union of their probe lists, loads
// each term2 row (and the first code chunk) ONCE, and runs one LUT-build + scan phase per
// query that probes the list.  Selection keys are (distance, position in the query's OWN
// probe order), a total order, so visiting the lists in union order changes nothing in the
// results -- they are bit-identical to scan16_kernel / the generic kernel.
#include "kernels.h"
#include "scan16_common.cuh"
#include "scan_common.cuh"
#include "wave_topk.cuh"

namespace vlq {

template <int KPL>
__global__ __launch_bounds__(256, (KPL <= 4 ? 4 : 2)) void scan16x2_kernel(ScanArgs a, int lut_region) {
    constexpr int E = 4096;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                         // [2][E]
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);              // [2 queries][4][64]
    unsigned char* p8 = reinterpret_cast<unsigned char*>(queue + 2 * 4 * 64);
    // two ProbeMeta blocks back to back; the one of query g is carved at pm_base + g * pm_stride
    // (computed, never an indexed array of structs: that would live in scratch memory)
    unsigned char* const pm_base = p8;
    const int pm_stride = (int)ProbeMeta::bytes(a.nprobe);
    ProbeMeta pm0, pm1;
    pm0.carve(pm_base, a.nprobe);
    pm1.carve(pm_base + pm_stride, a.nprobe);
    p8 += 2 * pm_stride;
    int32_t* matchA = reinterpret_cast<int32_t*>(p8);    // [nprobe] probe of B with the same list, or -1
    int32_t* matchB = matchA + a.nprobe;                 // [nprobe] probe of A with the same list, or -1
    uint32_t* phases = reinterpret_cast<uint32_t*>(matchB + a.nprobe);   // [2*nprobe]
    int32_t* misc = reinterpret_cast<int32_t*>(phases + 2 * a.nprobe);   // [0] number of phases

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // XCD-aware placement over PAIRS of consecutive entries of the sorted query order
    int64_t q0, q1;
    {
        const int64_t npairs = (a.nq + 1) >> 1;
        const int64_t b = blockIdx.x;
        const int64_t s = (b & 7) * a.xcd_chunk + (b >> 3);
        if (s >= npairs) return;
        q0 = a.qorder ? a.qorder[2 * s] : 2 * s;
        q1 = (2 * s + 1 < a.nq) ? (a.qorder ? a.qorder[2 * s + 1] : 2 * s + 1) : -1;
    }
    const bool haveB = q1 >= 0;

    // ---- per-query set-up: waves 0-1 gather A's probes, waves 2-3 B's ----------------
    bool badkey = false;
    if (t < 128) badkey = probe_meta_fill(a, q0, pm0, t, 128);
    else if (haveB) badkey = probe_meta_fill(a, q1, pm1, t - 128, 128);
    float4 m2t3A[4], m2t3B[4];
    load_query_table16(a, q0, t, lane, wave, m2t3A);
    if (haveB) load_query_table16(a, q1, t, lane, wave, m2t3B);
    __syncthreads();
    if (wave == 0) probe_meta_scan(a, pm0, lane);
    if (wave == 1 && haveB) probe_meta_scan(a, pm1, lane);
    __syncthreads();
    // which probes of the two queries name the same list (visited probes only)
    for (int j = t; j < a.nprobe; j += 256) { matchA[j] = -1; matchB[j] = -1; }
    __syncthreads();
    if (haveB) {
        for (int j = t; j < a.nprobe; j += 256) {
            const int32_t kb = pm1.pkey[j];
            if (kb < 0) continue;
            for (int i = 0; i < a.nprobe; i++)
                if (pm0.pkey[i] == kb) { matchB[j] = i; matchA[i] = j; break; }   // keys are distinct within a query
        }
    }
    __syncthreads();
    // phase list: A's visited probes in A's order, each followed by B's phase on the same
    // list if B probes it too, then B's remaining probes.  entry = probe | query << 16 |
    // same_list_as_previous_phase << 17
    if (t == 0) {
        int n = 0;
        for (int i = 0; i < a.nprobe; i++) {
            if (pm0.pkey[i] < 0) continue;
            phases[n++] = (uint32_t)i;
            if (matchA[i] >= 0) phases[n++] = (uint32_t)matchA[i] | (1u << 16) | (1u << 17);
        }
        if (haveB)
            for (int j = 0; j < a.nprobe; j++)
                if (pm1.pkey[j] >= 0 && matchB[j] < 0) phases[n++] = (uint32_t)j | (1u << 16);
        misc[0] = n;
    }
    __syncthreads();
    const int nph = misc[0];

    WaveSelect<KPL> selA, selB;
    selA.init(a.k, queue + wave * 64, lane);
    selB.init(a.k, queue + (4 + wave) * 64, lane);

    // ---- phase loop: term2 row + first code chunk fetched one LIST ahead ---------------
    float4 t2r[4];
    uint4 c0 = make_uint4(0, 0, 0, 0);
    auto fetch_list = [&](int ph) {     // ph = a phase that starts a new list
        const uint32_t e = phases[ph];
        ProbeMeta m;
        m.carve(pm_base + ((e >> 16) & 1) * pm_stride, a.nprobe);
        const int p = e & 0xffff;
        const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)m.pkey[p] * E);
#pragma unroll
        for (int i = 0; i < 4; i++) t2r[i] = src[i * 256 + t];
        if ((uint32_t)t < m.plen[p]) c0 = reinterpret_cast<const uint4*>(a.codes)[m.poff[p] + t];
    };
    if (nph > 0) fetch_list(0);
    uint64_t nscan = 0;
    for (int ph = 0; ph < nph; ph++) {
        const uint32_t e = phases[ph];
        const int g = (e >> 16) & 1;
        const int p = e & 0xffff;
        ProbeMeta m;
        m.carve(pm_base + g * pm_stride, a.nprobe);
        const uint32_t len = m.plen[p];
        const float dis0 = m.pd0[p];
        const uint32_t pos0 = m.cum[p];
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + m.poff[p];
        float* L = lut + (ph & 1) * E;
        if (g == 0) build_lut16(L, t, t2r, m2t3A);
        else build_lut16(L, t, t2r, m2t3B);
        uint4 cc = c0;
        // the row is still needed if the next phase is the other query on this list
        const bool next_same = (ph + 1 < nph) && ((phases[ph + 1] >> 17) & 1);
        if (!next_same && ph + 1 < nph) fetch_list(ph + 1);
        __syncthreads();
        for (uint32_t j0 = (uint32_t)wave * 64; j0 < len; j0 += 256) {
            const uint32_t j = j0 + lane;
            const uint32_t jn = j + 256;
            uint4 cn = make_uint4(0, 0, 0, 0);
            if (jn < len) cn = cp[jn];
            const bool valid = j < len;
            const float dis = adc16(L, cc, dis0);
            if (g == 0) selA.offer(dis, pos0 + j, valid);
            else selB.offer(dis, pos0 + j, valid);
            cc = cn;
        }
        nscan += len;
    }

    merge_and_emit<KPL>(selA, smraw, pm0.cum, a, q0, wave, lane,
                        [&](int pp, int64_t& lkey, int64_t& loff) { lkey = pm0.pkey[pp]; loff = pm0.poff[pp]; });
    if (haveB) {
        __syncthreads();   // merge area is reused
        merge_and_emit<KPL>(selB, smraw, pm1.cum, a, q1, wave, lane,
                            [&](int pp, int64_t& lkey, int64_t& loff) { lkey = pm1.pkey[pp]; loff = pm1.poff[pp]; });
    }
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
    if (badkey) *a.bad_key = 1;
}

template <int KPL>
static void launch_scan16x2_t(const ScanArgs& a, int lut_region, size_t smem, hipStream_t s) {
    static size_t attr_smem = 0;
    if (smem > attr_smem) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan16x2_kernel<KPL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_smem = smem;
    }
    const unsigned grid = (unsigned)(8 * a.xcd_chunk);
    hipLaunchKernelGGL((scan16x2_kernel<KPL>), dim3(grid), dim3(256), smem, s, a, lut_region);
}

void launch_scan16x2(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    const int64_t npairs = (a.nq + 1) / 2;
    a.xcd_chunk = (int)((npairs + 7) / 8);
    size_t lutb = (size_t)2 * 4096 * 4;
    const size_t merge = (size_t)4 * a.k * 8;
    if (lutb < merge) lutb = merge;
    const size_t tail = 2 * 4 * 64 * 8 + 2 * ((size_t)a.nprobe * 24 + 8) + (size_t)a.nprobe * 4 * 4 + 64;
    const size_t smem = lutb + tail;
    if (a.k <= 64) launch_scan16x2_t<1>(a, (int)lutb, smem, s);
    else if (a.k <= 256) launch_scan16x2_t<4>(a, (int)lutb, smem, s);
    else launch_scan16x2_t<16>(a, (int)lutb, smem, s);
}

}  // namespace vlq
