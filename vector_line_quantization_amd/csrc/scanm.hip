// List scan for 8-bit codes of every size from 4 to 64 bytes that is a multiple of 4, except 16 (M sub-quantizers x 256
// centroids, precomputed-table mode 1 or 2; mode 0 for 8- / 16- / 32-byte codes): the organisation of scan16_kernel
// (scan16.hip) over the code size.  The reference instantiates its scan per code size (gpu/impl/IVFPQ.cu:149-172,
// PQScanMultiPassPrecomputed.cu:1299-1313, loads in PQCodeLoad.cuh:60-357); its CPU scan is one loop over M
// (IndexIVFPQ.cpp:781-802).  Same arithmetic as the generic kernel in kernels.hip -- sim_table = term2[key] +
// (-2) * sim_table_2 (fvec_madd, IndexIVFPQ.cpp:641-644), dis = dis0 + tab[0][c0] + ... + tab[M-1][c_{M-1}] strictly left to
// right (:788-794) -- so results cannot depend on which kernel served a query; what changes is how the work is laid out:
//   * probe metadata gathered once per query into LDS, walking order without dead probes (ProbeMeta, scan16_common.cuh);
//   * the table has M x 256 entries = M KB: 4 waves per workgroup up to 32 bytes, 8 above (ScanMShape; 8-byte codes: 2 waves
//     from 3000 queries on): a thread owns M x 256 / threads table entries;
//   * (round 5) a list's chunks are requested a whole probe ahead into the registers the trip that consumed them has freed,
//     term2[key] behind the first chunk, every load of a probe unconditional and in one place (exact vmcnt counts); 8-byte codes
//     two probes ahead with two register sets; one workgroup barrier per probe with two table buffers (8-byte codes, mode 0),
//     two with one;
//   * gathers in half blocks of 8 sub-quantizers (a last one of 4 for 4-, 12-, 20-, 28-byte codes): one SDWA op per code byte
//     (byte extract and x4), sub-quantizer and buffer offsets in the ds_read offset field, two half blocks in flight while the
//     previous one is added;
//   * XCD-aware placement of the sorted query order, shared admission threshold of the workgroup's waves, keyed admission
//     (wave_topk.cuh: equal distances keep the reference's scan order).
#include <type_traits>

#include "kernels.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "walk_order.cuh"
#include "sse_order.cuh"
#include "wave_topk.cuh"

namespace vlq {

// one half block of 8 sub-quantizers: words w0, w1 against table slices at LDS byte OFFS + 0 .. 7 KB (no wait)
template <int OFFS>
__device__ __forceinline__ void issue_hb(float (&v)[8], uint32_t w0, uint32_t w1, uint32_t two) {
    VLQ_G8LO_NWI(OFFS, w0, w1);
}
template <int N>
__device__ __forceinline__ void wait_hb(float (&v)[8]) {
    if (N == 0) VLQ_WAIT8(0, v); else VLQ_WAIT8(8, v);
}
// a last half block of FOUR sub-quantizers (12-, 20-, 28-byte codes: one word against four table slices)
#define VLQ_G4_NWI(OFFS, W0)                                                                       \
    asm volatile(                                                                          \
        "v_lshlrev_b32_sdwa %0, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %1, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %2, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %3, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b32 %0, %0 offset:%6+0\n\t"                                               \
        "ds_read_b32 %1, %1 offset:%6+1024\n\t"                                            \
        "ds_read_b32 %2, %2 offset:%6+2048\n\t"                                            \
        "ds_read_b32 %3, %3 offset:%6+3072"                                                 \
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])                               \
        : "v"(W0), "v"(two), "n"(OFFS)                                                     \
        : "memory")
// sub-quantizers of half block S of an M-byte code, and the counted wait that says "half block S has arrived" while the next
// one (if any) is still in flight
template <int M> struct HbGeom {
    static constexpr int NH = (M + 7) / 8;
    static constexpr int size(int S) { return (S == NH - 1 && M % 8 == 4) ? 4 : 8; }
};
template <int M, int O, int S>
__device__ __forceinline__ void issue_hb_s(float (&v)[8], const uint32_t (&w)[M / 4], uint32_t two) {
    if constexpr (HbGeom<M>::size(S) == 8) issue_hb<O + S * 8192>(v, w[2 * S], w[2 * S + 1], two);
    else { VLQ_G4_NWI(O + S * 8192, w[2 * S]); }
}
template <int M, int S>
__device__ __forceinline__ void wait_add_s(float (&v)[8], float& dis) {
    constexpr int NH = HbGeom<M>::NH, N = HbGeom<M>::size(S);
    constexpr int BEHIND = S == NH - 1 ? 0 : HbGeom<M>::size(S + 1);       // reads still in flight behind this half block
    if constexpr (N == 8) {
        if constexpr (BEHIND == 0) VLQ_WAIT8(0, v); else if constexpr (BEHIND == 4) VLQ_WAIT8(4, v); else VLQ_WAIT8(8, v);
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
    }
#pragma unroll
    for (int m = 0; m < N; m++) dis = __fadd_rn(dis, v[m]);
    asm volatile("" : "+v"(dis));
}
template <int M, int O, int S>
__device__ __forceinline__ void adc_step(float (&hb)[HbGeom<M>::NH][8], const uint32_t (&w)[M / 4], float& dis, uint32_t two) {
    constexpr int NH = HbGeom<M>::NH;
    wait_add_s<M, S>(hb[S], dis);
    if constexpr (S + 2 < NH) issue_hb_s<M, O, S + 2>(hb[S + 2], w, two);
    if constexpr (S + 1 < NH) adc_step<M, O, S + 1>(hb, w, dis, two);
}

// dis + the M table values of one code, left to right; the table sits at LDS byte offset O (compile time), w = the code's words.
// Half blocks of 8 sub-quantizers (words 2h, 2h+1 against table slices 8h .. 8h+7; the last one of a 12-, 20- or 28-byte code
// has 4): two in flight, the adds of one under the reads of the next (counted lgkmcnt).
template <int M, int O>
__device__ __forceinline__ float adc_m(const uint32_t (&w)[M / 4], float dis, uint32_t two) {
    constexpr int NH = HbGeom<M>::NH;
    static_assert(O + (NH - 1) * 8192 + 7168 < 65536, "ds_read offsets are 16 bits");
    float hb[NH][8];
    issue_hb_s<M, O, 0>(hb[0], w, two);
    if constexpr (NH > 1) issue_hb_s<M, O, 1>(hb[1], w, two);
    adc_step<M, O, 0>(hb, w, dis, two);
    return dis;
}

// The same sum in two pieces: front() ends when the last half block's reads are issued -- every word of the code has been
// turned into addresses, its registers may be reloaded (the scan requests the next list's chunk there) -- back() adds the last
// two half blocks.  Same order of the additions, same counted waits.
template <int M, int O> struct AdcSplit {
    static constexpr int NH = HbGeom<M>::NH;
    float hb[NH][8];
    template <int S> __device__ __forceinline__ void add(float& dis) { wait_add_s<M, S>(hb[S], dis); }
    template <int S> __device__ __forceinline__ void roll(const uint32_t (&w)[M / 4], float& dis, uint32_t two) {
        if constexpr (S + 2 < NH) {
            add<S>(dis);
            issue_hb_s<M, O, S + 2>(hb[S + 2], w, two);
            roll<S + 1>(w, dis, two);
        }
    }
    __device__ __forceinline__ void front(const uint32_t (&w)[M / 4], float& dis, uint32_t two) {
        issue_hb_s<M, O, 0>(hb[0], w, two);
        if constexpr (NH > 1) issue_hb_s<M, O, 1>(hb[1], w, two);
        roll<0>(w, dis, two);
    }
    __device__ __forceinline__ void back(float& dis) {
        if constexpr (NH == 1) add<0>(dis);
        else { add<NH - 2>(dis); add<NH - 1>(dis); }
    }
};

template <int M> struct CodeWords { uint32_t w[M / 4]; };
template <int M>
__device__ __forceinline__ CodeWords<M> load_code(const uint8_t* __restrict__ base, int64_t row) {
    CodeWords<M> c;
    if constexpr (M % 16 == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(base) + row * (M / 16);
#pragma unroll
        for (int i = 0; i < M / 16; i++) {
            const uint4 v = p[i];
            c.w[4 * i] = v.x; c.w[4 * i + 1] = v.y; c.w[4 * i + 2] = v.z; c.w[4 * i + 3] = v.w;
        }
    } else if constexpr (M % 8 == 0) {      // 8, 24, 40, 56 bytes: rows are 8-byte aligned
        const uint2* p = reinterpret_cast<const uint2*>(base) + row * (M / 8);
#pragma unroll
        for (int i = 0; i < M / 8; i++) {
            const uint2 v = p[i];
            c.w[2 * i] = v.x; c.w[2 * i + 1] = v.y;
        }
    } else {                                // 12, 20, 28 bytes: 4-byte aligned
        const uint32_t* p = reinterpret_cast<const uint32_t*>(base) + row * (M / 4);
#pragma unroll
        for (int i = 0; i < M / 4; i++) c.w[i] = p[i];
    }
    return c;
}

template <int M> struct ScanMShape {
    // waves per workgroup: 4 up to 32 bytes, 8 for 64.
    // 32-byte codes ran 8 waves x 16 entries per thread with two 32 KB buffers (2 workgroups per CU) until late round 4; 4 waves
    // x 32 entries with ONE buffer (128 VGPRs forced, 4 workgroups per CU) halves the lane slots a ~330-code list leaves empty
    // (256-wide trips instead of 512) -- 10 000 queries: headline data 1.36 -> 1.24 ms, G1 1.37 -> 1.16 (the LDS gather rate of
    // tools/micro/lds_gather.hip), k = 100 1.75 -> 1.62 / 1.21, k = 200 2.5 -> 1.66; a 1250-query slice loses 8 % (0.222 -> 0.242).
    // 8-byte codes with 2 waves measured slower (0.371 -> 0.388).  64-byte codes with 16 waves and 16 entries per thread
    // measured 3.92 / 4.25 ms on the two bench data sets against 3.82 / 4.00 with 8 waves and 32 entries per thread -- 145
    // VGPRs, one workgroup per CU either way; forcing 128 VGPRs for two workgroups spills: 4.41 / 3.81 (round 5, with the
    // chunks requested a probe ahead: 3.62 / 3.77 at one workgroup, 4.05 / 3.34 forced to two -- the headline data's 64 KB rows,
    // 21 GB requested per launch, are what bounds it: more workgroups in flight only lower the L2 hit rate).
    static constexpr int NW = M <= 32 ? 4 : 8;
    static constexpr int NT = 64 * NW;
    static constexpr int E = M * 256;
    static constexpr int NI = E / 4 / NT;                    // float4 of the table per thread: 2 (M = 8), 4, or 8 (M = 32, 64)
};

// DSUB > 0: table mode 0 (by_residual WITHOUT the precomputed table -- GpuIndexIVFPQConfig::usePrecomputedTables = false, the
// reference GPU class's default; IndexIVFPQ.cpp:636-637 on the CPU): the table of a (query, list) pair is
// compute_distance_table(x - centroid) = |(x - c)_m - cent_mj|^2 in fvec_L2sqr's order (utils.cpp:481-506), dis0 = 0.  A thread
// keeps the DSUB components of its 4 * NI centroids in registers for the whole query (d floats: 128 VGPRs at d = 128), the
// residual of the NEXT probe is formed one probe ahead in LDS; the generic kernel re-read the 128 KB codebook from L2 per
// probe (3.25 ms per 10 000 queries on the bench index against 0.73 with the precomputed table).
// NWX: waves per workgroup when not the shape's (8-byte codes, large batches: 2 -- see launch_scanm_k)
template <int M, int KPL, int NBUF, bool IMI, int DSUB = 0, int NWX = 0>
__global__ __launch_bounds__(64 * (NWX ? NWX : ScanMShape<M>::NW)) __attribute__((amdgpu_waves_per_eu((M == 32 && DSUB == 0 && KPL == 1) ? 4 : (M == 8 && DSUB == 0 && KPL == 1) ? (NWX == 2 ? 5 : 6) : 1))) void scanm_kernel(ScanArgs a, int lut_region) {
    constexpr int NW = NWX ? NWX : ScanMShape<M>::NW, NT = 64 * NW, E = ScanMShape<M>::E, NI = E / 4 / NT;
    static_assert(DSUB == 0 || (NBUF == 2 && !IMI), "table mode 0: two table buffers, flat coarse quantizer");
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                         // [NBUF][E] at LDS byte 0
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);              // [NW][64]
    ProbeMeta pm;
    pm.carve(reinterpret_cast<unsigned char*>(queue + NW * 64), a.nprobe);
    int32_t* misc = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(queue + NW * 64) + ProbeMeta::bytes(a.nprobe));
    uint16_t* ord = reinterpret_cast<uint16_t*>(misc + 2);                      // [nprobe] visited probes, in walking order
    uint32_t* wg_thr = reinterpret_cast<uint32_t*>(ord + ((a.nprobe + 1) & ~1));
    float* sres = reinterpret_cast<float*>(wg_thr + 4);                          // [2][M * DSUB] residuals (table mode 0)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }    // adc_m addresses the buffers from LDS byte 0
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
    // XCD-aware placement (scan16.hip): XCD x serves the x-th contiguous chunk of the sorted query order; small batches split
    // a query's probes over a.nsplit workgroups writing partial rows
    int64_t q;
    int part = 0;
    {
        const int64_t b = blockIdx.x;
        const int64_t s = (b & 7) * a.xcd_chunk + (b >> 3);
        if (s >= a.nq * a.nsplit) return;
        const int64_t qs = s / a.nsplit;
        part = (int)(s - qs * a.nsplit);
        q = a.qorder ? a.qorder[qs] : qs;
    }
    const int64_t* kq = a.keys + q * a.nprobe;

    const bool badkey = probe_meta_fill(a, q, pm, t, NT);
    // -2 * sim_table_2 of the query, entries 4*(i*NT+t) .. +3: sub-quantizer NW*i + wave, centroids 4*lane .. +3
    float4 m2t3[NI];
    float cbk[DSUB > 0 ? NI * 4 * DSUB : 1];        // table mode 0: the thread's centroids, entry (i, c) = centroid 4*lane + c of m = NW*i + wave
    float qmine = 0.f;                              // ... and component t of the query
    if constexpr (DSUB > 0) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const float4* src = reinterpret_cast<const float4*>(a.pq_cent + ((size_t)(NW * i + wave) * 256 + 4 * lane) * DSUB);
#pragma unroll
            for (int v = 0; v < DSUB; v++) {        // 4 * DSUB contiguous floats
                const float4 f = src[v];
                cbk[i * 4 * DSUB + 4 * v] = f.x; cbk[i * 4 * DSUB + 4 * v + 1] = f.y;
                cbk[i * 4 * DSUB + 4 * v + 2] = f.z; cbk[i * 4 * DSUB + 4 * v + 3] = f.w;
            }
        }
        if (t < M * DSUB) qmine = a.queries[q * (M * DSUB) + t];
    } else if (a.qtab) {                            // materialised by launch_pq_tables
        const float4* qt = reinterpret_cast<const float4*>(a.qtab + q * E);
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const float4 v = qt[i * NT + t];
            m2t3[i] = make_float4(__fmul_rn(-2.f, v.x), __fmul_rn(-2.f, v.y), __fmul_rn(-2.f, v.z), __fmul_rn(-2.f, v.w));
        }
    } else {
        // ProductQuantizer::compute_inner_prod_table (ProductQuantizer.cpp:424-436) in fvec_inner_product's order
        // (utils.cpp:509-533: lane accumulator c % 4 takes component c in increasing order, the zero-padded tail and the
        // unconditional + 0 of the last block, then (s0 + s1) + (s2 + s3)) from the transposed codebook [m][component][j]:
        // the four centroids of a thread are one 16-byte load per component, a wave reads 1 KiB contiguous
        // (one slice at a time, parked in the still unused table buffer: unrolled over the NI slices the 16 accumulators and
        // their loads set the kernel's register count -- 131 instead of 100 VGPRs at M = 32, a workgroup less per CU)
        const float* qv = a.queries + q * a.d;
        const int dsub = a.dsub;
#pragma unroll 1
        for (int i = 0; i < NI; i++) {
            const int m = NW * i + wave;
            const float4* ct = reinterpret_cast<const float4*>(a.pq_cent_t + (size_t)m * dsub * 256) + lane;
            const float* xm = qv + m * dsub;
            float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
            auto madd = [](float4& s, float x, const float4 y) __attribute__((always_inline)) {
                s.x = __fadd_rn(s.x, __fmul_rn(x, y.x)); s.y = __fadd_rn(s.y, __fmul_rn(x, y.y));
                s.z = __fadd_rn(s.z, __fmul_rn(x, y.z)); s.w = __fadd_rn(s.w, __fmul_rn(x, y.w));
            };
            auto add0 = [](float4& s) __attribute__((always_inline)) {
                s.x = __fadd_rn(s.x, 0.f); s.y = __fadd_rn(s.y, 0.f); s.z = __fadd_rn(s.z, 0.f); s.w = __fadd_rn(s.w, 0.f);
            };
            int c = 0;
            for (; c + 4 <= dsub; c += 4) {
                madd(s0, xm[c], ct[(c + 0) * 64]); madd(s1, xm[c + 1], ct[(c + 1) * 64]);
                madd(s2, xm[c + 2], ct[(c + 2) * 64]); madd(s3, xm[c + 3], ct[(c + 3) * 64]);
            }
            const int r = dsub - c;
            if (r > 0) madd(s0, xm[c], ct[c * 64]); else add0(s0);
            if (r > 1) madd(s1, xm[c + 1], ct[(c + 1) * 64]); else add0(s1);
            if (r > 2) madd(s2, xm[c + 2], ct[(c + 2) * 64]); else add0(s2);
            add0(s3);
            reinterpret_cast<float4*>(lut)[i * NT + t] =
                make_float4(__fmul_rn(-2.f, __fadd_rn(__fadd_rn(s0.x, s1.x), __fadd_rn(s2.x, s3.x))),
                            __fmul_rn(-2.f, __fadd_rn(__fadd_rn(s0.y, s1.y), __fadd_rn(s2.y, s3.y))),
                            __fmul_rn(-2.f, __fadd_rn(__fadd_rn(s0.z, s1.z), __fadd_rn(s2.z, s3.z))),
                            __fmul_rn(-2.f, __fadd_rn(__fadd_rn(s0.w, s1.w), __fadd_rn(s2.w, s3.w))));
        }
#pragma unroll
        for (int i = 0; i < NI; i++) m2t3[i] = reinterpret_cast<const float4*>(lut)[i * NT + t];   // the thread's own stores
    }
    __syncthreads();
    int walk_mean = -1;        // thread 0: walk_order.cuh
    if (wave == 0) {
        const int cut = probe_meta_scan(a, pm, lane);
        __builtin_amdgcn_wave_barrier();
        int nl = 0;
        for (int p0 = 0; p0 < cut; p0 += 64) {      // coarse-distance order, dead probes dropped
            const int p = p0 + lane;
            const bool lv = p < cut && pm.pkey[p] >= 0;
            const u64 mask = __ballot(lv);
            if (lv) ord[nl + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)p;
            nl += __popcll(mask);
        }
        if (a.nsplit == 1) walk_mean = walk_order_sort(a, pm, ord, nl, lane);    // parts are merged in part order = scan order
        if (lane == 0) { misc[0] = cut; misc[1] = nl; *wg_thr = f32_to_ordered(3.402823466e+38f); }
    }
    __syncthreads();
    const int nlive = misc[1];

    WaveSelect<KPL, 1, (KPL >= 2)> sel;
    sel.init(a.k, queue + wave * 64, lane);

    CodeWords<M> c0;
#pragma unroll
    for (int i = 0; i < M / 4; i++) c0.w[i] = 0;
    // (scan16.hip, round 5) the codes of a list are requested a whole probe ahead, chunk c of the next list into the registers
    // the trip that consumed chunk c of this one has just freed; every load of a probe is issued unconditionally and in one
    // place, so the compiler's vmcnt counts are exact (with the next chunk requested under a condition inside the list loop it
    // waited for vmcnt(0) in every trip: for the NEXT probe's row and codes, requested just before the barrier).
    // 8-byte codes: TWO probes ahead (DEPTH 2, register sets alternating with the table buffers): three trips of 8 gathers
    // are over long before a row requested one probe earlier has crossed the fabric.
    #ifdef VLQ_SCANM8_OLD
    constexpr bool AHEAD = DSUB == 0 && KPL <= 2 && M == 32;
#else
    constexpr bool AHEAD = DSUB == 0 && KPL <= 2;
#endif
    constexpr int DEPTH = (AHEAD && NBUF == 2) ? 2 : 1;
    constexpr int NPRE = M == 8 ? 3 : 2;           // chunks of a list requested ahead (32-byte codes: registers)
    constexpr int NEX = M == 8 ? 4 : 1;            // chunks of a longer list in flight at a time
    struct ProbeRegs { uint32_t len = 0, pos0 = 0; float dis0 = 0.f; int64_t off = 0, key = 0; };   // (wave-uniform)
    ProbeRegs nx[DEPTH];
    float4 t2s[DEPTH][NI];
    CodeWords<M> cr[DEPTH][NPRE];
#pragma unroll
    for (int s = 0; s < DEPTH; s++)
#pragma unroll
        for (int c = 0; c < NPRE; c++)
#pragma unroll
            for (int i = 0; i < M / 4; i++) cr[s][c].w[i] = 0;
    auto prefetch_meta = [&](int i, auto set_) __attribute__((always_inline)) {
        constexpr int S = decltype(set_)::value;
        if (i >= nlive) return;      // (a part may look past its range, the last probes past the end: harmless repeated loads)
        const int p = ord[i];
        nx[S].key = (int64_t)__builtin_amdgcn_readfirstlane(pm.pkey[p]);
        nx[S].len = __builtin_amdgcn_readfirstlane(pm.plen[p]);
        nx[S].dis0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pm.pd0[p])));
        nx[S].pos0 = __builtin_amdgcn_readfirstlane(pm.cum[p]);
        const int64_t o = pm.poff[p];
        nx[S].off = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)o >> 32)) << 32) |
                              __builtin_amdgcn_readfirstlane((uint32_t)o));
    };
    auto load_rows = [&](auto set_) __attribute__((always_inline)) {
        constexpr int S = decltype(set_)::value;
        const int64_t key = nx[S].key;
        if constexpr (DSUB > 0) {
            // (table mode 0: no term2 row; the coarse centroid is requested by fetch_coarse, two probes ahead)
        } else if (IMI) {
            // table type 2 (IndexIVFPQ.cpp:645-686): sub-quantizer m = NW*i + wave takes its 1 KB slice from the row of the
            // coarse sub-index of its half
            const int64_t ki0 = key & ((int64_t(1) << a.imi_nbits) - 1), ki1 = key >> a.imi_nbits;
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) {
                const int64_t ki = (NW * i2 + wave) < M / 2 ? ki0 : ki1;
                t2s[S][i2] = reinterpret_cast<const float4*>(a.term2 + (size_t)ki * E)[i2 * NT + t];
            }
        } else {
            const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)key * E);
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) t2s[S][i2] = src[i2 * NT + t];
        }
    };
    auto load_chunk = [&](auto set_, auto cc_) __attribute__((always_inline)) {
        constexpr int S = decltype(set_)::value, C = decltype(cc_)::value;
        cr[S][C] = load_code<M>(a.codes, nx[S].off + (int64_t)min((uint32_t)t + C * NT, nx[S].len - 1));
    };
    // (guarded: the first request of a workgroup, which may have no live probe at all; inside the probe loop the loads are
    // unconditional -- past the last probe they repeat its addresses -- a load under a condition made the compiler copy half a
    // row behind a vmcnt(6) right after requesting it: 64-byte codes 3.82 -> 4.18 ms)
    auto prefetch = [&](int i, auto set_, bool guarded) __attribute__((always_inline)) {
        if (guarded && i >= nlive) return;
        prefetch_meta(i, set_);
        load_rows(set_);
        if (AHEAD) {
            load_chunk(set_, std::integral_constant<int, 0>{});
            load_chunk(set_, std::integral_constant<int, 1>{});
            if constexpr (NPRE == 3) load_chunk(set_, std::integral_constant<int, 2>{});
        } else c0 = load_code<M>(a.codes, nx[0].off + (int64_t)min((uint32_t)t, nx[0].len - 1));
    };
    const int i_begin = (int)((int64_t)part * nlive / a.nsplit), i_end = (int)((int64_t)(part + 1) * nlive / a.nsplit);
    const unsigned long long t_walk = wall_clock64();
    prefetch(i_begin, std::integral_constant<int, 0>{}, true);
    if constexpr (DEPTH == 2) prefetch(i_begin + 1, std::integral_constant<int, 1>{}, true);
    if (AHEAD) __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): nothing pending on the way into the loop (scan16.hip)
    // table mode 0: component t of the coarse centroid of the i-th walked probe, requested two probes ahead; the residual
    // x - centroid (compute_residual, IndexIVFPQ.cpp:636) of the next probe is written to LDS while the current table is built
    constexpr int DV = M * (DSUB > 0 ? DSUB : 1);
    float cnext = 0.f;
    int rb = 0;
    auto fetch_coarse = [&](int i) __attribute__((always_inline)) {
        if (DSUB == 0 || i >= nlive || t >= DV) return;
        cnext = a.coarse[(size_t)pm.pkey[ord[i]] * DV + t];
    };
    if constexpr (DSUB > 0) {
        fetch_coarse(i_begin);
        if (t < DV) sres[t] = __fsub_rn(qmine, cnext);
        fetch_coarse(i_begin + 1);
        __syncthreads();
    }
    int buf = 0;
    uint64_t nscan = 0;
    // one probe; BUF >= 0: the table buffer is known at compile time (the pair loop below), -1: taken from buf
    auto probe = [&](int i, auto fixed_) __attribute__((always_inline)) {
        constexpr int FIXED = decltype(fixed_)::value;
        constexpr int S = DEPTH == 2 ? FIXED : 0;     // register set of this probe
        static_assert(DEPTH == 1 || FIXED >= 0, "two probes ahead: the pair loop");
        const std::integral_constant<int, S> set_{};
        const uint32_t len = nx[S].len, pos0 = nx[S].pos0;
        const float dis0 = DSUB > 0 ? 0.f : nx[S].dis0;  // mode 0: dis0 = 0 (IndexIVFPQ.cpp:638)
        const int64_t off = nx[S].off;
        float* L = lut + (FIXED >= 0 ? FIXED : buf) * E;
        if (NBUF == 1) __syncthreads();              // single table buffer: everyone is done scanning with it
        __builtin_amdgcn_s_setprio(2);               // table build + the next list's first loads first (scan16.hip)
        if constexpr (DSUB > 0) {
            if (t < DV && i + 1 < nlive) sres[(rb ^ 1) * DV + t] = __fsub_rn(qmine, cnext);
            fetch_coarse(i + 2);
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) {
                const float* rs = sres + rb * DV + (NW * i2 + wave) * DSUB;      // wave-uniform: LDS broadcast reads
                float r[DSUB];
#pragma unroll
                for (int c2 = 0; c2 < DSUB; c2++) r[c2] = rs[c2];
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float* cj = cbk + (i2 * 4 + c) * DSUB;
                    v[c] = l2sqr_sse_order([&](int c2) { return r[c2]; }, [&](int c2) { return cj[c2]; }, DSUB);
                }
                reinterpret_cast<float4*>(L)[i2 * NT + t] = make_float4(v[0], v[1], v[2], v[3]);
            }
            rb ^= 1;
        } else {
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) {
                float4 sv;
                sv.x = __fadd_rn(t2s[S][i2].x, m2t3[i2].x); sv.y = __fadd_rn(t2s[S][i2].y, m2t3[i2].y);
                sv.z = __fadd_rn(t2s[S][i2].z, m2t3[i2].z); sv.w = __fadd_rn(t2s[S][i2].w, m2t3[i2].w);
                reinterpret_cast<float4*>(L)[i2 * NT + t] = sv;
            }
        }
        CodeWords<M> cc = c0;
        if (AHEAD) prefetch_meta(i + DEPTH, set_); else prefetch(i + 1, set_, false);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (sel.dirty) {
            if (lane == 0) atomicMin(wg_thr, f32_to_ordered(sel.thr_own));
            sel.dirty = false;
        }
        sel.refresh_with(*wg_thr);
        auto scan_list = [&](auto bufc) {
            constexpr int B = decltype(bufc)::value;
            uint32_t j0 = (uint32_t)wave * 64;
            if constexpr (AHEAD) {
                const uint32_t w64 = j0;
                // a list longer than NPRE chunks: its further chunks first (the order inside a list is free: walk_order.cuh), NEX
                // at a time, reloaded in place (clamped: behind the list's end a cache hit nobody uses).  Requested here and not
                // before the barrier: registers defined under one condition and used under another made the compiler copy them
                // behind a vmcnt(0) on EVERY path.
                const uint32_t w64x = w64 + NPRE * NT;
                if (w64x < len) {
                    CodeWords<M> ex[NEX];
#pragma unroll
                    for (int e = 0; e < NEX; e++) ex[e] = load_code<M>(a.codes, off + (int64_t)min(w64x + e * NT + lane, len - 1));
                    uint32_t jx = w64x;
                    for (; jx + NEX * NT < len; jx += NEX * NT) {        // rounds with a further round behind them: all NEX chunks real
#pragma unroll
                        for (int e = 0; e < NEX; e++) {
                            const uint32_t j0e = jx + e * NT;
                            AdcSplit<M, B * E * 4> ad;
                            float dis = dis0;
                            ad.front(ex[e].w, dis, two);
                            ex[e] = load_code<M>(a.codes, off + (int64_t)min(j0e + NEX * NT + lane, len - 1));
                            ad.back(dis);
                            sel.offer_keyed(dis, pos0 + j0e + lane, j0e + lane < len);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < NEX; e++) {                      // the last round requests nothing
                        const uint32_t j0e = jx + e * NT;
                        if (j0e < len) {
                            const float dis = adc_m<M, B * E * 4>(ex[e].w, dis0, two);
                            sel.offer_keyed(dis, pos0 + j0e + lane, j0e + lane < len);
                        }
                    }
                    // (nothing of this block is in flight behind it: what the compiler must assume pending where the two
                    // paths meet decides the waits of EVERY probe)
                    if (NEX > 1) __builtin_amdgcn_s_waitcnt(0x0F70);
                }
                auto trip = [&](auto cc_) {
                    constexpr int C = decltype(cc_)::value;
                    const uint32_t jc = w64 + C * NT;
                    uint32_t g = jc < len ? 1u : 0u;      // (wave-uniform)
                    AdcSplit<M, B * E * 4> ad;
                    float dis = dis0;
                    if (g) ad.front(cr[S][C].w, dis, two);
                    load_chunk(set_, cc_);
                    if (C == 0) load_rows(set_);      // behind the chunk: vmcnt retires in order
                    g = __builtin_amdgcn_readfirstlane(g);
                    asm volatile("" : "+s"(g));        // (keeps the two halves of the trip from being threaded into two copies of the loads)
                    if (g) {
                        ad.back(dis);
                        sel.offer_keyed(dis, pos0 + jc + lane, jc + lane < len);
                    }
                };
                {
                    // (8-byte codes: chunks 0 and 1 as ONE trip -- the adds of the first under the reads of the second, as the
                    // loop without the look-ahead has it -- measured 0.50 / 0.54 ms on the two bench data sets against 0.36 /
                    // 0.42: its 16 more live registers do not fit the 80 of six waves per SIMD)
                    trip(std::integral_constant<int, 0>{});
                    trip(std::integral_constant<int, 1>{});
                    if constexpr (NPRE == 3) trip(std::integral_constant<int, 2>{});
                }
                return;
            }
            if constexpr (M == 8) {
                // 8-byte codes are one half block: two chunks per trip, the adds of the first under the reads of the second
                // (what the half-block pipeline does inside a longer code)
                for (; j0 + NT < len; j0 += 2 * NT) {
                    const uint32_t ja = j0 + lane, jb = ja + NT;
                    const CodeWords<M> cb = load_code<M>(a.codes, off + (int64_t)min(jb, len - 1));
                    CodeWords<M> cn = cc;
                    if (j0 + 2 * NT < len) cn = load_code<M>(a.codes, off + (int64_t)min(jb + NT, len - 1));
                    float ha[8], hb2[8];
                    issue_hb<B * E * 4>(ha, cc.w[0], cc.w[1], two);
                    issue_hb<B * E * 4>(hb2, cb.w[0], cb.w[1], two);
                    wait_hb<8>(ha);
                    float da = dis0;
#pragma unroll
                    for (int m = 0; m < 8; m++) da = __fadd_rn(da, ha[m]);
                    asm volatile("" : "+v"(da));
                    wait_hb<0>(hb2);
                    float db = dis0;
#pragma unroll
                    for (int m = 0; m < 8; m++) db = __fadd_rn(db, hb2[m]);
                    sel.offer_keyed(da, pos0 + ja, true);
                    sel.offer_keyed(db, pos0 + jb, jb < len);
                    cc = cn;
                }
            }
            for (; j0 < len; j0 += NT) {
                const uint32_t j = j0 + lane;
                CodeWords<M> cn = cc;
                if (j0 + NT < len) cn = load_code<M>(a.codes, off + (int64_t)min(j + NT, len - 1));   // (wave-uniform)
                const float dis = adc_m<M, B * E * 4>(cc.w, dis0, two);
                sel.offer_keyed(dis, pos0 + j, j < len);
                cc = cn;
            }
        };
        if constexpr (FIXED >= 0) scan_list(std::integral_constant<int, FIXED>{});
        else if (NBUF == 1 || buf == 0) scan_list(std::integral_constant<int, 0>{});
        else scan_list(std::integral_constant<int, NBUF == 2 ? 1 : 0>{});
        nscan += len;
        if (NBUF == 2 && FIXED < 0) buf ^= 1;
    };
    if constexpr (AHEAD && NBUF == 2) {
        // two probes per trip of the loop, one per table buffer: with one copy of the list loop per buffer behind a run-time
        // choice, the registers the next list's codes are loaded into had two load sites merging at the loop's end -- copies
        // behind a vmcnt(0), i.e. a wait for everything just requested
        int i = i_begin;
        for (; i + 1 < i_end; i += 2) {
            probe(i, std::integral_constant<int, 0>{});
            probe(i + 1, std::integral_constant<int, 1>{});
        }
        if (i < i_end) probe(i, std::integral_constant<int, 0>{});
    } else {
        for (int i = i_begin; i < i_end; i++) probe(i, std::integral_constant<int, -1>{});
    }
    if (t == 0) walk_state_finish(a, t_walk, i_end - i_begin, walk_mean);
    merge_and_emit<KPL, NW>(sel, smraw, pm.cum, a, a.nsplit > 1 ? (int64_t)part * a.nq + q : q, wave, lane,
                            [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
    if (badkey) *a.bad_key = 1;
}

template <int M, int KPL, int NBUF, bool IMI, int DSUB = 0, int NWX = 0>
static void launch_scanm_i(const ScanArgs& a, hipStream_t s) {
    constexpr int NW = NWX ? NWX : ScanMShape<M>::NW, E = ScanMShape<M>::E;
    size_t lutb = (size_t)NBUF * E * 4;
    const size_t merge = (size_t)NW * a.k * 8;
    if (lutb < merge) lutb = merge;
    const size_t tail = (size_t)NW * 64 * 8 + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 2 + 8 + 64 + (size_t)2 * M * DSUB * 4 + 16;
    const size_t smem = lutb + tail;
    ensure_dynamic_lds(reinterpret_cast<const void*>(scanm_kernel<M, KPL, NBUF, IMI, DSUB, NWX>), smem);
    hipLaunchKernelGGL((scanm_kernel<M, KPL, NBUF, IMI, DSUB, NWX>), dim3((unsigned)(8 * a.xcd_chunk)), dim3(64 * NW), smem, s, a, (int)lutb);
}
// table mode 0 (DSUB = d / M components per sub-quantizer)
template <int M, int DSUB>
static void launch_scanm0_k(const ScanArgs& a, hipStream_t s) {
    if (a.k <= 64) launch_scanm_i<M, 1, 2, false, DSUB>(a, s);
    else if (a.k <= 256) launch_scanm_i<M, 4, 2, false, DSUB>(a, s);
    else launch_scanm_i<M, 16, 2, false, DSUB>(a, s);
}
template <int M, int NBUF>
static void launch_scanm_k(const ScanArgs& a, hipStream_t s) {
#define VLQ_SM(K)                                                       \
    do {                                                                \
        if (a.imi_nbits > 0) launch_scanm_i<M, K, NBUF, true>(a, s);    \
        else launch_scanm_i<M, K, NBUF, false>(a, s);                   \
    } while (0)
    // 8-byte codes, k <= 64, 3000 queries and more: two waves per workgroup.  The kernel is bound by the instructions it issues
    // (profiles/r05_code_sizes.txt: 4100 VALU + 2400 SALU per wave at four waves, 3/4 of them per-probe work every wave
    // repeats -- metadata, addresses, table build, threshold -- for 1.3 trips of 8 gathers); two waves halve that share
    static const int nw8_env = [] { const char* e = getenv("VLQ_SCANM8_WAVES"); return e ? atoi(e) : 0; }();
    if (M == 8 && a.k <= 64 && (nw8_env ? nw8_env == 2 : a.nq * a.nsplit >= 3000)) {
        if (a.imi_nbits > 0) launch_scanm_i<M, 1, NBUF, true, 0, M == 8 ? 2 : 0>(a, s);
        else launch_scanm_i<M, 1, NBUF, false, 0, M == 8 ? 2 : 0>(a, s);
    } else if (a.k <= 64) VLQ_SM(1);
    else if (a.k <= 128) VLQ_SM(2);
    else if (a.k <= 256) VLQ_SM(4);
    else VLQ_SM(16);
#undef VLQ_SM
}

// table mode 0 on the engineered kernel: 8-, 16- and 32-byte codes, flat coarse quantizer, d <= 128 (a thread holds d codebook floats)
bool scanm0_supports(const ScanArgs& a) {
    if (!(a.table_mode == 0 && a.ksub == 256 && a.imi_nbits == 0 && a.coarse && a.pq_cent && a.queries && a.nprobe <= 1024)) return false;
    if (a.M == 16) return a.dsub == 4 || a.dsub == 6 || a.dsub == 8;
    if (a.M == 8) return a.dsub == 8 || a.dsub == 12 || a.dsub == 16;
    if (a.M == 32) return a.dsub == 2 || a.dsub == 4;         // (round 5: d = 64 / 128; 64-byte codes would put the second table
    return false;                                             //  buffer past the 16-bit offset of the gather instructions)
}

bool scanm_supports(const ScanArgs& a) {
    return (a.M == 4 || a.M == 8 || a.M == 12 || (a.M >= 20 && a.M <= 32 && a.M % 4 == 0) || (a.M >= 40 && a.M <= 64 && a.M % 8 == 0)) &&
           a.ksub == 256 && a.table_mode == 1 && (a.qtab || a.pq_cent_t) && a.term2 &&
           a.nprobe <= 1024 &&
           (a.imi_nbits == 0 || a.M % 2 == 0);
}

void launch_scanm(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    if (a.nsplit < 1) a.nsplit = 1;
    a.xcd_chunk = (int)((a.nq * a.nsplit + 7) / 8);
    if (a.table_mode == 0) {
        if (a.M == 16) {
            if (a.dsub == 8) launch_scanm0_k<16, 8>(a, s);
            else if (a.dsub == 6) launch_scanm0_k<16, 6>(a, s);
            else launch_scanm0_k<16, 4>(a, s);
        } else if (a.M == 32) {
            if (a.dsub == 4) launch_scanm0_k<32, 4>(a, s);
            else launch_scanm0_k<32, 2>(a, s);
        } else {
            if (a.dsub == 16) launch_scanm0_k<8, 16>(a, s);
            else if (a.dsub == 12) launch_scanm0_k<8, 12>(a, s);
            else launch_scanm0_k<8, 8>(a, s);
        }
        return;
    }
    // (round 5: 4-, 12-, 20-, 24-, 28-, 40-, 48- and 56-byte codes -- the other multiples of 4 bytes that the reference
    // instantiates, gpu/impl/IVFPQ.cu:149-172 -- are the same template: ceil(M / 8) half blocks, the last one of 4 where M is
    // not a multiple of 8, one table buffer, 4 waves up to 32 bytes, 8 above)
    switch (a.M) {
    case 8: launch_scanm_k<8, 2>(a, s); break;
    case 4: launch_scanm_k<4, 1>(a, s); break;
    case 12: launch_scanm_k<12, 1>(a, s); break;
    case 20: launch_scanm_k<20, 1>(a, s); break;
    case 24: launch_scanm_k<24, 1>(a, s); break;
    case 28: launch_scanm_k<28, 1>(a, s); break;
    case 32: launch_scanm_k<32, 1>(a, s); break;      // one 32 KB buffer, 4 waves (ScanMShape)
    case 40: launch_scanm_k<40, 1>(a, s); break;
    case 48: launch_scanm_k<48, 1>(a, s); break;
    case 56: launch_scanm_k<56, 1>(a, s); break;
    default: launch_scanm_k<64, 1>(a, s); break;
    }
}

void preload_scanm_kernels() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(scanm_kernel<8, 1, 2, false, 0, 0>));
}

}  // namespace vlq
