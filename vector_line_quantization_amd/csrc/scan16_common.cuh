// Pieces shared by the 16-byte-code scan kernels (scan16.hip, scan16x2.hip): identical
// arithmetic in both, so results cannot depend on which kernel served a query.
#pragma once
#include "kernels.h"
#include "wave_topk.cuh"

namespace vlq {

// -2 * sim_table_2 of query q (ProductQuantizer::compute_inner_prod_table,
// ProductQuantizer.cpp:424-436), 16 entries per thread.  Entry e = 4*(i*256+t)+c ->
// sub-quantizer m = NWV*i + wave (wave-uniform), centroid j = 4*lane + c.
template <int NI>   // NI float4 per thread: 4 with 256 threads, 2 with 512
__device__ __forceinline__ void load_query_table16(const ScanArgs& a, int64_t q, int t, int lane,
                                                   int wave, float4 (&m2t3)[NI]) {
    constexpr int E = 4096;
    constexpr int NT = 1024 / NI;     // threads per workgroup
    constexpr int NWV = NT / 64;      // waves per workgroup
    if (a.qtab) {
        const float4* qt = reinterpret_cast<const float4*>(a.qtab + q * E);
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const float4 v = qt[i * NT + t];
            m2t3[i] = a.qtab_scaled ? v
                                    : make_float4(__fmul_rn(-2.f, v.x), __fmul_rn(-2.f, v.y), __fmul_rn(-2.f, v.z),
                                                  __fmul_rn(-2.f, v.w));
        }
    } else {
        // codebook read from its transposed copy pq_cent_t[m][component][j]: the four
        // centroids j = 4*lane..4*lane+3 of one component are one 16-byte load, a wave
        // reads 1 KiB contiguous per instruction
        const float* qv = a.queries + q * 128;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int m = NWV * i + wave;
            const float4* ct = reinterpret_cast<const float4*>(a.pq_cent_t + (size_t)m * 8 * 256) + lane;
            const float4 x0 = *reinterpret_cast<const float4*>(qv + m * 8);
            const float4 x1 = *reinterpret_cast<const float4*>(qv + m * 8 + 4);
            const float4 y0 = ct[0 * 64], y1 = ct[1 * 64], y2 = ct[2 * 64], y3 = ct[3 * 64];
            const float4 y4 = ct[4 * 64], y5 = ct[5 * 64], y6 = ct[6 * 64], y7 = ct[7 * 64];
            // fvec_inner_product, d = 8 (utils.cpp:509-533): s_l = ((0 + x_l y_l) + x_{l+4} y_{l+4}) + 0,
            // result (s0+s1)+(s2+s3); .x/.y/.z/.w = centroids 4*lane+0..3
#define VLQ_IP8(C)                                                                                        \
    __fmul_rn(-2.f,                                                                                      \
              __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.x, y0.C)), __fmul_rn(x1.x, y4.C)), 0.f), \
                                  __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.y, y1.C)), __fmul_rn(x1.y, y5.C)), 0.f)), \
                        __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.z, y2.C)), __fmul_rn(x1.z, y6.C)), 0.f), \
                                  __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.w, y3.C)), __fmul_rn(x1.w, y7.C)), 0.f))))
            m2t3[i] = make_float4(VLQ_IP8(x), VLQ_IP8(y), VLQ_IP8(z), VLQ_IP8(w));
#undef VLQ_IP8
        }
    }
}

// dis = dis0 + tab[0][c0] + ... + tab[15][c15], strictly left to right
// (IndexIVFPQ.cpp:788-794); L = LUT [16][256] in LDS, cc = one 16-byte code
__device__ __forceinline__ float adc16(const float* L, const uint4 cc, float dis) {
    dis = __fadd_rn(dis, L[0 * 256 + (cc.x & 255u)]);
    dis = __fadd_rn(dis, L[1 * 256 + ((cc.x >> 8) & 255u)]);
    dis = __fadd_rn(dis, L[2 * 256 + ((cc.x >> 16) & 255u)]);
    dis = __fadd_rn(dis, L[3 * 256 + (cc.x >> 24)]);
    dis = __fadd_rn(dis, L[4 * 256 + (cc.y & 255u)]);
    dis = __fadd_rn(dis, L[5 * 256 + ((cc.y >> 8) & 255u)]);
    dis = __fadd_rn(dis, L[6 * 256 + ((cc.y >> 16) & 255u)]);
    dis = __fadd_rn(dis, L[7 * 256 + (cc.y >> 24)]);
    dis = __fadd_rn(dis, L[8 * 256 + (cc.z & 255u)]);
    dis = __fadd_rn(dis, L[9 * 256 + ((cc.z >> 8) & 255u)]);
    dis = __fadd_rn(dis, L[10 * 256 + ((cc.z >> 16) & 255u)]);
    dis = __fadd_rn(dis, L[11 * 256 + (cc.z >> 24)]);
    dis = __fadd_rn(dis, L[12 * 256 + (cc.w & 255u)]);
    dis = __fadd_rn(dis, L[13 * 256 + ((cc.w >> 8) & 255u)]);
    dis = __fadd_rn(dis, L[14 * 256 + ((cc.w >> 16) & 255u)]);
    dis = __fadd_rn(dis, L[15 * 256 + (cc.w >> 24)]);
    return dis;
}

// Hand-scheduled form of the 16 LUT gathers of one code for a LUT at a FIXED LDS byte
// offset (0 or 16384: the two buffers at the start of the kernel's dynamic LDS, which
// starts at LDS address 0 because the kernel declares no static LDS -- checked at kernel
// entry).  hipcc spends two VALU ops per lookup (v_bfe_u32 + v_lshl_add_u32); SDWA does the
// byte extract and the x4 in one, and m*1024 + buffer offset ride in the ds_read offset
// field.  All 16 reads are issued back to back, one wait, then the caller adds left to
// right -- the same values and the same addition order as adc16().
#define VLQ_G16_ASM(O)                                                                              \
    asm volatile(                                                                                   \
        "v_lshlrev_b32_sdwa %0, %20, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"  \
        "v_lshlrev_b32_sdwa %1, %20, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"  \
        "v_lshlrev_b32_sdwa %2, %20, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"  \
        "v_lshlrev_b32_sdwa %3, %20, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"  \
        "v_lshlrev_b32_sdwa %4, %20, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"  \
        "v_lshlrev_b32_sdwa %5, %20, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"  \
        "v_lshlrev_b32_sdwa %6, %20, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"  \
        "v_lshlrev_b32_sdwa %7, %20, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"  \
        "v_lshlrev_b32_sdwa %8, %20, %18 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"  \
        "v_lshlrev_b32_sdwa %9, %20, %18 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"  \
        "v_lshlrev_b32_sdwa %10, %20, %18 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %11, %20, %18 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "v_lshlrev_b32_sdwa %12, %20, %19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %13, %20, %19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %14, %20, %19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %15, %20, %19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b32 %0, %0 offset:" #O "+0\n\t"                                                      \
        "ds_read_b32 %1, %1 offset:" #O "+1024\n\t"                                                   \
        "ds_read_b32 %2, %2 offset:" #O "+2048\n\t"                                                   \
        "ds_read_b32 %3, %3 offset:" #O "+3072\n\t"                                                   \
        "ds_read_b32 %4, %4 offset:" #O "+4096\n\t"                                                   \
        "ds_read_b32 %5, %5 offset:" #O "+5120\n\t"                                                   \
        "ds_read_b32 %6, %6 offset:" #O "+6144\n\t"                                                   \
        "ds_read_b32 %7, %7 offset:" #O "+7168\n\t"                                                   \
        "ds_read_b32 %8, %8 offset:" #O "+8192\n\t"                                                   \
        "ds_read_b32 %9, %9 offset:" #O "+9216\n\t"                                                   \
        "ds_read_b32 %10, %10 offset:" #O "+10240\n\t"                                                \
        "ds_read_b32 %11, %11 offset:" #O "+11264\n\t"                                                \
        "ds_read_b32 %12, %12 offset:" #O "+12288\n\t"                                                \
        "ds_read_b32 %13, %13 offset:" #O "+13312\n\t"                                                \
        "ds_read_b32 %14, %14 offset:" #O "+14336\n\t"                                                \
        "ds_read_b32 %15, %15 offset:" #O "+15360\n\t"                                                \
        "s_waitcnt lgkmcnt(0)"                                                                      \
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]),  \
          "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]),            \
          "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])                                                   \
        : "v"(cc.x), "v"(cc.y), "v"(cc.z), "v"(cc.w), "v"(two)                                       \
        : "memory")

// Half blocks of VLQ_G16_ASM (sub-quantizers 0-7 from code words x,y; 8-15 from z,w) WITHOUT the
// trailing wait, and a counted wait: lgkmcnt is a 4-bit counter, so a wave can have 16 LDS reads in
// flight -- two half blocks.  The list loop keeps one half block in flight while it adds the other.
#define VLQ_G8LO_NW(O, W0, W1)                                                                   \
    asm volatile(                                                                          \
        "v_lshlrev_b32_sdwa %0, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %1, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %2, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %3, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "v_lshlrev_b32_sdwa %4, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %5, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %6, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %7, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b32 %0, %0 offset:" #O "+0\n\t" \
        "ds_read_b32 %1, %1 offset:" #O "+1024\n\t" \
        "ds_read_b32 %2, %2 offset:" #O "+2048\n\t" \
        "ds_read_b32 %3, %3 offset:" #O "+3072\n\t" \
        "ds_read_b32 %4, %4 offset:" #O "+4096\n\t" \
        "ds_read_b32 %5, %5 offset:" #O "+5120\n\t" \
        "ds_read_b32 %6, %6 offset:" #O "+6144\n\t" \
        "ds_read_b32 %7, %7 offset:" #O "+7168"                                                                                 \
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])                                                                               \
        : "v"(W0), "v"(W1), "v"(two)                                                       \
        : "memory")
#define VLQ_G8HI_NW(O, W0, W1)                                                                   \
    asm volatile(                                                                          \
        "v_lshlrev_b32_sdwa %0, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %1, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %2, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %3, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "v_lshlrev_b32_sdwa %4, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %5, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %6, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %7, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b32 %0, %0 offset:" #O "+8192\n\t" \
        "ds_read_b32 %1, %1 offset:" #O "+9216\n\t" \
        "ds_read_b32 %2, %2 offset:" #O "+10240\n\t" \
        "ds_read_b32 %3, %3 offset:" #O "+11264\n\t" \
        "ds_read_b32 %4, %4 offset:" #O "+12288\n\t" \
        "ds_read_b32 %5, %5 offset:" #O "+13312\n\t" \
        "ds_read_b32 %6, %6 offset:" #O "+14336\n\t" \
        "ds_read_b32 %7, %7 offset:" #O "+15360"                                                                                 \
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])                                                                               \
        : "v"(W0), "v"(W1), "v"(two)                                                       \
        : "memory")
#define VLQ_WAIT8(N, v)                                                                             \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                        \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                 :: "memory")

// the same half blocks with the LUT offset as a compile-time operand (template parameters)
#define VLQ_G8LO_NWI(OFFS, W0, W1)                                                                \
    asm volatile(                                                                          \
        "v_lshlrev_b32_sdwa %0, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %1, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %2, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %3, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "v_lshlrev_b32_sdwa %4, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %5, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %6, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %7, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b32 %0, %0 offset:%11+0\n\t" \
        "ds_read_b32 %1, %1 offset:%11+1024\n\t" \
        "ds_read_b32 %2, %2 offset:%11+2048\n\t" \
        "ds_read_b32 %3, %3 offset:%11+3072\n\t" \
        "ds_read_b32 %4, %4 offset:%11+4096\n\t" \
        "ds_read_b32 %5, %5 offset:%11+5120\n\t" \
        "ds_read_b32 %6, %6 offset:%11+6144\n\t" \
        "ds_read_b32 %7, %7 offset:%11+7168"                                                                                 \
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])                                                                               \
        : "v"(W0), "v"(W1), "v"(two), "n"(OFFS)                                            \
        : "memory")
#define VLQ_G8HI_NWI(OFFS, W0, W1)                                                                \
    asm volatile(                                                                          \
        "v_lshlrev_b32_sdwa %0, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %1, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %2, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %3, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "v_lshlrev_b32_sdwa %4, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %5, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %6, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %7, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b32 %0, %0 offset:%11+8192\n\t" \
        "ds_read_b32 %1, %1 offset:%11+9216\n\t" \
        "ds_read_b32 %2, %2 offset:%11+10240\n\t" \
        "ds_read_b32 %3, %3 offset:%11+11264\n\t" \
        "ds_read_b32 %4, %4 offset:%11+12288\n\t" \
        "ds_read_b32 %5, %5 offset:%11+13312\n\t" \
        "ds_read_b32 %6, %6 offset:%11+14336\n\t" \
        "ds_read_b32 %7, %7 offset:%11+15360"                                                                                 \
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])                                                                               \
        : "v"(W0), "v"(W1), "v"(two), "n"(OFFS)                                            \
        : "memory")
// N chunks against the LUT at LDS byte offset O as one straight-line half-block pipeline: 16 reads
// in flight at the start, then always 8 in flight under the 8 dependent adds of the previous half
// block; acc[c] = dis0 + the 16 table values of chunk c, left to right.  No control flow inside.
template <int N, int O>
__device__ __forceinline__ void adc16_pipeline(const uint4 (&c)[N], float dis0, uint32_t two, float (&acc)[N]) {
    float hb[2 * N][8];
#define VLQ_ISSUE(S)                                                                                    \
    do {                                                                                                \
        float (&v)[8] = hb[S];                                                                          \
        if ((S) % 2 == 0) VLQ_G8LO_NWI(O, c[(S) / 2].x, c[(S) / 2].y);                                   \
        else VLQ_G8HI_NWI(O, c[(S) / 2].z, c[(S) / 2].w);                                                \
    } while (0)
    VLQ_ISSUE(0);
    VLQ_ISSUE(1);
#pragma unroll
    for (int s = 0; s < 2 * N; s++) {
        if (s == 2 * N - 1) VLQ_WAIT8(0, hb[s]); else VLQ_WAIT8(8, hb[s]);
        float d = (s % 2 == 0) ? dis0 : acc[s / 2];
#pragma unroll
        for (int m = 0; m < 8; m++) d = __fadd_rn(d, hb[s][m]);
        asm volatile("" : "+v"(d));
        acc[s / 2] = d;
        if (s + 2 < 2 * N) VLQ_ISSUE(s + 2);
    }
#undef VLQ_ISSUE
}

// one chunk in two half blocks: the 8 adds of the first half run while the second half's reads are in flight
template <int BUF>
__device__ __forceinline__ float adc16_halves(const uint4 cc, float dis, uint32_t two) {
    float lo[8], hi[8];
    if (BUF == 0) { { float (&v)[8] = lo; VLQ_G8LO_NW(0, cc.x, cc.y); } { float (&v)[8] = hi; VLQ_G8HI_NW(0, cc.z, cc.w); } }
    else { { float (&v)[8] = lo; VLQ_G8LO_NW(16384, cc.x, cc.y); } { float (&v)[8] = hi; VLQ_G8HI_NW(16384, cc.z, cc.w); } }
    VLQ_WAIT8(8, lo);
#pragma unroll
    for (int m = 0; m < 8; m++) dis = __fadd_rn(dis, lo[m]);
    asm volatile("" : "+v"(dis));
    VLQ_WAIT8(0, hi);
#pragma unroll
    for (int m = 0; m < 8; m++) dis = __fadd_rn(dis, hi[m]);
    return dis;
}

// the same, with `after_issue()` run once both half blocks are issued and the code's registers are dead: a reload of THOSE
// registers placed there needs no second register set (and no copy at a loop's back edge, whose wait would be for the reload)
template <int BUF, typename F>
__device__ __forceinline__ float adc16_halves_then(const uint4 cc, float dis, uint32_t two, F&& after_issue) {
    float lo[8], hi[8];
    if (BUF == 0) { { float (&v)[8] = lo; VLQ_G8LO_NW(0, cc.x, cc.y); } { float (&v)[8] = hi; VLQ_G8HI_NW(0, cc.z, cc.w); } }
    else { { float (&v)[8] = lo; VLQ_G8LO_NW(16384, cc.x, cc.y); } { float (&v)[8] = hi; VLQ_G8HI_NW(16384, cc.z, cc.w); } }
    after_issue();
    VLQ_WAIT8(8, lo);
#pragma unroll
    for (int m = 0; m < 8; m++) dis = __fadd_rn(dis, lo[m]);
    asm volatile("" : "+v"(dis));
    VLQ_WAIT8(0, hi);
#pragma unroll
    for (int m = 0; m < 8; m++) dis = __fadd_rn(dis, hi[m]);
    return dis;
}

template <int BUF>
__device__ __forceinline__ float adc16_fixed(const uint4 cc, float dis, uint32_t two) {
    float v[16];
    if (BUF == 0) { VLQ_G16_ASM(0); } else { VLQ_G16_ASM(16384); }
#pragma unroll
    for (int m = 0; m < 16; m++) dis = __fadd_rn(dis, v[m]);
    return dis;
}

// sim_table = term2[key] + (-2) * sim_table_2  (fvec_madd, IndexIVFPQ.cpp:641-644):
// 4*NI entries per thread, NI 16-byte LDS stores
template <int NI>
__device__ __forceinline__ void build_lut16(float* L, int t, const float4 (&t2)[NI],
                                            const float4 (&m2t3)[NI]) {
    constexpr int NT = 1024 / NI;
#pragma unroll
    for (int i = 0; i < NI; i++) {
        float4 s;
        s.x = __fadd_rn(t2[i].x, m2t3[i].x);
        s.y = __fadd_rn(t2[i].y, m2t3[i].y);
        s.z = __fadd_rn(t2[i].z, m2t3[i].z);
        s.w = __fadd_rn(t2[i].w, m2t3[i].w);
        reinterpret_cast<float4*>(L)[i * NT + t] = s;
    }
}

// Per-query probe metadata in LDS.
struct ProbeMeta {
    int64_t* poff;    // [nprobe] list start (codes/ids row)
    uint32_t* cum;    // [nprobe+1] scan position of the probe's first code
    uint32_t* plen;   // [nprobe]
    int32_t* pkey;    // [nprobe] list id, -1 = not visited (invalid key, empty list, behind max_codes)
    float* pd0;       // [nprobe] coarse distance
    __device__ __forceinline__ static size_t bytes(int nprobe) { return (size_t)nprobe * 24 + 8; }
    __device__ __forceinline__ void carve(unsigned char* base, int nprobe) {
        poff = reinterpret_cast<int64_t*>(base);
        cum = reinterpret_cast<uint32_t*>(poff + nprobe);
        plen = cum + nprobe + 1;
        pkey = reinterpret_cast<int32_t*>(plen + nprobe);
        pd0 = reinterpret_cast<float*>(pkey + nprobe);
    }
};

// step 1 (all threads, stride nthr starting at t0): gather keys / offsets / lengths
__device__ __forceinline__ bool probe_meta_fill(const ScanArgs& a, int64_t q, ProbeMeta& pm, int t0,
                                                int nthr) {
    const int64_t* kq = a.keys + q * a.nprobe;
    const float* cq = a.coarse_dis + q * a.nprobe;
    bool badkey = false;
    for (int p = t0; p < a.nprobe; p += nthr) {
        const int64_t key = kq[p];
        if (key >= a.nlist) badkey = true;                 // IndexIVFPQ.cpp:1008-1011
        const bool live = key >= 0 && key < a.nlist;
        int64_t off = 0, len = 0;
        if (live) { off = a.list_off[key]; len = a.list_len ? a.list_len[key] : a.list_off[key + 1] - off; }
        pm.poff[p] = off;
        pm.plen[p] = (uint32_t)len;
        pm.pkey[p] = (live && len > 0) ? (int32_t)key : -1;   // empty lists are skipped (:1016)
        pm.pd0[p] = cq[p];
    }
    return badkey;
}

// step 2 (ONE wave, after a barrier): exclusive prefix sum of the list lengths and the
// max_codes cut (IndexIVFPQ.cpp:1033: stop after the probe that reaches it); probes behind
// the cut are marked not visited.  Returns the number of probes visited.
__device__ __forceinline__ int probe_meta_scan(const ScanArgs& a, ProbeMeta& pm, int lane) {
    if (a.nprobe <= 64) {
        // one probe per lane, VALU only (round 5): the cross-lane steps of the general form below are ds_bpermutes, and
        // under the list scans' gathers each waits hundreds of cycles in the LDS queue -- 18 of them sat between the set-up
        // barriers of every workgroup.  Lengths are summed as 26 + 6 bit halves (64 x 2^32 < 2^38).
        const bool in = lane < a.nprobe;
        const uint32_t len = in ? pm.plen[lane] : 0u;
        const uint32_t lo = wave_scan_incl_u32(len & 0x3ffffffu), hi = wave_scan_incl_u32(len >> 26);
        const uint64_t incl = ((uint64_t)hi << 26) + lo;
        if (in) pm.cum[lane] = (uint32_t)(incl - len);
        const u64 over = __ballot(in && a.max_codes && incl >= (uint64_t)a.max_codes);
        const int cut = over ? __builtin_ctzll(over) + 1 : a.nprobe;         // first probe index AFTER the cut
        if (lane == 63) pm.cum[a.nprobe] = (uint32_t)incl;
        __builtin_amdgcn_wave_barrier();
        if (cut < a.nprobe) {
            const uint32_t endpos = pm.cum[cut];
            __builtin_amdgcn_wave_barrier();
            for (int p = cut + lane; p <= a.nprobe; p += 64) pm.cum[p] = endpos;
            for (int p = cut + lane; p < a.nprobe; p += 64) pm.pkey[p] = -1;
        }
        return cut;
    }
    const int per = (a.nprobe + 63) >> 6;
    const int p0 = lane * per;
    uint64_t local = 0;
    for (int i = 0; i < per; i++) { const int p = p0 + i; if (p < a.nprobe) local += pm.plen[p]; }
    uint64_t incl = local;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)incl, sft, 64);
        const uint32_t hi = __shfl_up((uint32_t)(incl >> 32), sft, 64);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        if (lane >= sft) incl += o;
    }
    uint64_t run = incl - local;
    int cut = a.nprobe;                                  // first probe index AFTER the cut
    for (int i = 0; i < per; i++) {
        const int p = p0 + i;
        if (p < a.nprobe) {
            pm.cum[p] = (uint32_t)run;
            run += pm.plen[p];
            if (a.max_codes && run >= (uint64_t)a.max_codes && cut == a.nprobe) cut = p + 1;
        }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) cut = min(cut, __shfl_xor(cut, sft, 64));
    if (lane == 63) pm.cum[a.nprobe] = (uint32_t)incl;
    __builtin_amdgcn_wave_barrier();
    if (cut < a.nprobe) {
        const uint32_t endpos = pm.cum[cut];
        __builtin_amdgcn_wave_barrier();
        for (int p = cut + lane; p <= a.nprobe; p += 64) pm.cum[p] = endpos;
        for (int p = cut + lane; p < a.nprobe; p += 64) pm.pkey[p] = -1;
    }
    return cut;
}

}  // namespace vlq
