// Wave-autonomous list scan for 16-byte codes (M = 16, ksub = 256, table mode 1 / 2).
// Same arithmetic as scan16.hip (IndexIVFPQ.cpp:631-690, :781-802); what changes is who
// synchronises with whom:
//   * a workgroup (NW waves) still serves one query, but every WAVE owns its own 16 KB LUT
//     in LDS and claims whole probes from an LDS counter.  Building a LUT (16 ds_write_b128
//     per lane) and scanning the list with it are ordered by the LDS pipe itself (DS
//     instructions of one wave execute in issue order), so the probe loop has NO workgroup
//     barrier: the CU's LDS pipe always has NW x (workgroups per CU) independent streams;
//   * the whole per-query table part (-2 * sim_table_2, 16 KB) is held by every wave in
//     64 VGPRs; it is computed once per workgroup and exchanged through LDS;
//   * term2[key] (64 VGPRs) and the first code chunk are prefetched one claimed probe ahead.
// Results are bit-identical to scan16.hip and to the generic kernel: the selection key
// (distance, scan position) is a total order, so it does not matter which wave scans what.
#include <type_traits>

#include "kernels.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "wave_topk.cuh"

namespace vlq {

#ifdef VLQ_STAMPS
// kernel experiments only (tools/stamps.py): per-phase shader-clock sums over all waves
__device__ unsigned long long g_stamps[16];
#define STAMP() __builtin_amdgcn_s_memtime()
#define ACC(i, v) do { if (lane == 0) atomicAdd(&g_stamps[i], (unsigned long long)(v)); } while (0)
#else
#define STAMP() 0ull
#define ACC(i, v) do { } while (0)
#endif

// gathers of one code against the LUT of wave W (LDS byte offset W * 16384)
template <int W>
__device__ __forceinline__ float adc16_wave(const uint4 cc, float dis, uint32_t two) {
    float v[16];
    if (W == 0) { VLQ_G16_ASM(0); }
    else if (W == 1) { VLQ_G16_ASM(16384); }
    else if (W == 2) { VLQ_G16_ASM(32768); }
    else { VLQ_G16_ASM(49152); }
#pragma unroll
    for (int m = 0; m < 16; m++) dis = __fadd_rn(dis, v[m]);
    return dis;
}

template <int KPL, int NW>
__global__ __launch_bounds__(64 * NW) void scan16w_kernel(ScanArgs a) {
    constexpr int PD = 4;             // code chunks (64 codes each) kept in flight per wave
    constexpr int E = 4096;
    constexpr int NT = 64 * NW;
    constexpr int NI = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float4* lut4 = reinterpret_cast<float4*>(smraw);                         // [NW][16][64] float4
    u64* queue = reinterpret_cast<u64*>(smraw + (size_t)NW * E * 4);         // [NW][64]
    ProbeMeta pm;
    pm.carve(reinterpret_cast<unsigned char*>(queue + NW * 64), a.nprobe);
    int32_t* misc = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(queue + NW * 64) +
                                               ProbeMeta::bytes(a.nprobe));   // cut, nlive, counter, pad
    uint16_t* live = reinterpret_cast<uint16_t*>(misc + 4);                  // [nprobe] visited probes, in order

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }   // adc16_wave: LUTs at LDS 0..
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
    const unsigned long long st0 = STAMP();

    // ---- per-query set-up ---------------------------------------------------------
    const bool badkey = probe_meta_fill(a, q, pm, t, NT);
    float4 qt[16];
    {
        float4 part[NI];
        load_query_table16<NI>(a, q, t, lane, wave, part);     // sub-quantizers NW*i + wave
#pragma unroll
        for (int i = 0; i < NI; i++) lut4[(NW * i + wave) * 64 + lane] = part[i];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; m++) qt[m] = lut4[m * 64 + lane];
    if (wave == 0) {
        const int cut = probe_meta_scan(a, pm, lane);
        __builtin_amdgcn_wave_barrier();
        int nl = 0;
        for (int p0 = 0; p0 < cut; p0 += 64) {
            const int p = p0 + lane;
            const bool lv = p < cut && pm.pkey[p] >= 0;
            const u64 mask = __ballot(lv);
            if (lv) live[nl + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)p;
            nl += __popcll(mask);
        }
        if (lane == 0) { misc[0] = cut; misc[1] = nl; misc[2] = 0; }
    }
    __syncthreads();     // also: every wave has its copy of qt before wave 0's LUT is rebuilt
    const int nlive = misc[1];

    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);
    float4* L = lut4 + wave * (E / 4);

    auto claim = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(&misc[2], 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    float4 t2r[16];
    uint4 cnx[PD];
#pragma unroll
    for (int u = 0; u < PD; u++) cnx[u] = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](int i) {
        if (i >= nlive) return;
        const int p = live[i];
        const int64_t key = pm.pkey[p];
        if (a.imi_nbits > 0) {     // table type 2 (IndexIVFPQ.cpp:645-686)
            const int64_t ki0 = key & ((int64_t(1) << a.imi_nbits) - 1), ki1 = key >> a.imi_nbits;
            const float4* s0 = reinterpret_cast<const float4*>(a.term2 + (size_t)ki0 * E) + lane;
            const float4* s1 = reinterpret_cast<const float4*>(a.term2 + (size_t)ki1 * E) + lane;
#pragma unroll
            for (int m = 0; m < 16; m++) t2r[m] = (m < 8 ? s0 : s1)[m * 64];
        } else {
#if !(defined(VLQ_ABL) && (VLQ_ABL & 8))
            const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)key * E) + lane;
#pragma unroll
            for (int m = 0; m < 16; m++) t2r[m] = src[m * 64];
#endif
        }
        // the first PD chunks of the list, clamped instead of predicated (branch-free loads)
        const uint4* cpn = reinterpret_cast<const uint4*>(a.codes) + pm.poff[p];
        const uint32_t last = pm.plen[p] - 1;
#pragma unroll
        for (int u = 0; u < PD; u++) cnx[u] = cpn[min((uint32_t)(u * 64 + lane), last)];
    };

    int cur = claim();
    prefetch(cur);
    const unsigned long long st1 = STAMP();
    ACC(0, st1 - st0);
    unsigned long long s_x[3] = {0, 0, 0};
    unsigned long long s_wait = 0, s_build = 0, s_pref = 0, s_scan = 0, s_iter = 0;
    while (cur < nlive) {
        const unsigned long long sa = STAMP();
#ifdef VLQ_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned long long sb = STAMP();
        const int p = live[cur];
        const uint32_t len = pm.plen[p];
        const float dis0 = pm.pd0[p];
        const uint32_t pos0 = pm.cum[p];
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + pm.poff[p];
        // sim_table = term2[key] + (-2) * sim_table_2 (fvec_madd, IndexIVFPQ.cpp:641-644)
#pragma unroll
        for (int m = 0; m < 16; m++) {
            float4 s;
            s.x = __fadd_rn(t2r[m].x, qt[m].x);
            s.y = __fadd_rn(t2r[m].y, qt[m].y);
            s.z = __fadd_rn(t2r[m].z, qt[m].z);
            s.w = __fadd_rn(t2r[m].w, qt[m].w);
#if defined(VLQ_ABL) && (VLQ_ABL & 2)
            asm volatile("" :: "v"(s.x), "v"(s.y), "v"(s.z), "v"(s.w));
#else
            L[m * 64 + lane] = s;
#endif
        }
        uint4 cq[PD];
#pragma unroll
        for (int u = 0; u < PD; u++) cq[u] = cnx[u];
#ifdef VLQ_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        const unsigned long long sc = STAMP();
        const int nxt = claim();
        prefetch(nxt);
        const unsigned long long sd = STAMP();
        const uint32_t nchunk = (len + 63) >> 6;
        // one copy of the list loop per wave: the LUT's LDS offset is an immediate
        auto scan_list = [&](auto wc) {
            constexpr int W = decltype(wc)::value;
            // half-block software pipeline (adc16_pipeline) over up to PD chunks per trip
            for (uint32_t i0 = 0; i0 < nchunk; i0 += PD) {
                uint4 c[PD];
#pragma unroll
                for (int u = 0; u < PD; u++) {
                    c[u] = cq[u];
                    cq[u] = cp[min((i0 + u + PD) * 64 + lane, len - 1)];       // chunks PD ahead, clamped, unconditional
                }
                const uint32_t left = nchunk - i0;
                float acc[PD];
                if (left >= PD) {
                    adc16_pipeline<PD, W * 16384>(c, dis0, two, acc);
                } else if (left >= 2) {
                    const uint4 c2[2] = {c[0], c[1]};
                    float a2[2];
                    adc16_pipeline<2, W * 16384>(c2, dis0, two, a2);
                    acc[0] = a2[0]; acc[1] = a2[1];
                    if (left == 3) acc[2] = adc16_wave<W>(c[2], dis0, two);
                } else {
                    acc[0] = adc16_wave<W>(c[0], dis0, two);
                }
#pragma unroll
                for (int u = 0; u < PD; u++) {
                    const uint32_t j = (i0 + u) * 64 + lane;
                    const bool in = (uint32_t)u < left && j < len;
                    if (__builtin_amdgcn_ballot_w64(in && acc[u] < sel.thr)) sel.offer(acc[u], pos0 + j, in);
                }
            }
        };
        if (wave == 0) scan_list(std::integral_constant<int, 0>{});
        else if (wave == 1) scan_list(std::integral_constant<int, 1>{});
        else if (wave == 2) scan_list(std::integral_constant<int, 2>{});
        else scan_list(std::integral_constant<int, 3>{});
        cur = nxt;
        const unsigned long long se = STAMP();
        s_wait += sb - sa; s_build += sc - sb; s_pref += sd - sc; s_scan += se - sd; s_iter += (len + 63) >> 6;
    }
    const unsigned long long st2 = STAMP();
    ACC(1, s_wait); ACC(2, s_build); ACC(3, s_pref); ACC(4, s_scan); ACC(5, s_iter); ACC(6, st2 - st1); ACC(8, 1); ACC(9, s_x[0]); ACC(10, s_x[1]); ACC(11, s_x[2]);

    merge_and_emit<KPL, NW>(sel, smraw, pm.cum, a, q, wave, lane,
                            [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)pm.cum[a.nprobe]);
    if (badkey) *a.bad_key = 1;
    ACC(7, STAMP() - st2);
}

#ifdef VLQ_STAMPS
extern "C" int vlq_debug_stamps(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

template <int KPL, int NW>
static void launch_scan16w_t(const ScanArgs& a, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan16w_kernel<KPL, NW>), smem);
    const unsigned grid = (unsigned)(8 * a.xcd_chunk);
    hipLaunchKernelGGL((scan16w_kernel<KPL, NW>), dim3(grid), dim3(64 * NW), smem, s, a);
}

void launch_scan16w(const ScanArgs& a_in, int nw, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    a.xcd_chunk = (int)((a.nq + 7) / 8);
    const size_t smem = (size_t)nw * 16384 + (size_t)nw * 64 * 8 + (size_t)a.nprobe * 24 + 8 + 16 +
                        (size_t)a.nprobe * 2 + 64;
    if (nw == 1) {
        if (a.k <= 64) launch_scan16w_t<1, 1>(a, smem, s);
        else if (a.k <= 256) launch_scan16w_t<4, 1>(a, smem, s);
        else launch_scan16w_t<16, 1>(a, smem, s);
    } else if (nw == 2) {
        if (a.k <= 64) launch_scan16w_t<1, 2>(a, smem, s);
        else if (a.k <= 256) launch_scan16w_t<4, 2>(a, smem, s);
        else launch_scan16w_t<16, 2>(a, smem, s);
    } else {
        if (a.k <= 64) launch_scan16w_t<1, 4>(a, smem, s);
        else if (a.k <= 256) launch_scan16w_t<4, 4>(a, smem, s);
        else launch_scan16w_t<16, 4>(a, smem, s);
    }
}

}  // namespace vlq
