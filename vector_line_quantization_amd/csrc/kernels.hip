// HIP kernels of the IVFPQ search path for gfx950 (MI355X, CDNA4).
//
// Stage map (reference function -> kernel):
//   fvec_norms_L2sqr            utils.cpp:675-682        -> row_norms_kernel
//   knn_L2sqr_blas (sgemm part) utils.cpp:834-901        -> coarse_dist_kernel   (f32 MFMA)
//   knn_L2sqr_blas (heap part)  utils.cpp:876-893        -> coarse_select_kernel (wave select)
//   compute_inner_prod_table    ProductQuantizer.cpp:424 -> pq_tables_kernel
//   precompute_table            IndexIVFPQ.cpp:392-429   -> pq_tables_kernel (mode 2)
//   precompute_list_tables_L2 + scan_list_with_table + heap
//                               IndexIVFPQ.cpp:631-690, :781-802, :964-1060
//                                                        -> scan_kernel
//   compute_code                ProductQuantizer.cpp:311-336 -> encode_kernel
#include "kernels.h"
#include <type_traits>

#include <map>
#include <mutex>
#include <utility>
#include "sse_order.cuh"
#include "wave_topk.cuh"
#include "scan_common.cuh"

namespace vlq {

void ensure_dynamic_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> high;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t& h = high[std::make_pair(kernel, dev)];
    if (bytes > h) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        h = bytes;
    }
}

#define FLT_MAX_F 3.402823466e+38f

// ---------------------------------------------------------------------------
// row norms
// ---------------------------------------------------------------------------
__global__ void row_norms_kernel(const float* __restrict__ x, int64_t n, int d,
                                 float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* xi = x + i * d;
    out[i] = norm_sse_order([&](int c) { return xi[c]; }, d);
}

void preload_search_kernels() {
    static bool done = false;           // (per process: the code objects stay loaded)
    if (done) return;
    done = true;
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(row_norms_kernel));
    preload_scan16_kernels();
    preload_coarse_screen_kernels();
    preload_scanm_kernels();
    (void)hipGetLastError();
}

void launch_row_norms(const float* x, int64_t n, int d, float* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n,
                       d, out);
}

// ---------------------------------------------------------------------------
// coarse distances: 128x128 output tile per 256-thread workgroup, 4 waves as 2x2,
// each wave a 64x64 sub-tile = 2x2 v_mfma_f32_32x32x2_f32 accumulators.
// The MFMA consumes k in pairs (lane>>5 selects k parity), so the LDS tiles are
// stored de-interleaved (even k | odd k): every lane then reads 4 consecutive
// k-pairs with one ds_read_b128 and the accumulation runs over k = 0,1,2,... in
// natural order -- bit-identical to a scalar fmaf chain.
// Row stride KC/2+4 floats keeps the b128 reads bank-conflict free.
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KC>
__global__ __launch_bounds__(256) void coarse_dist_kernel(
    const float* __restrict__ Q, const float* __restrict__ Cn, const float* __restrict__ qn,
    const float* __restrict__ cn, float* __restrict__ out, int64_t nq, int nlist, int d) {
    constexpr int S = KC / 2 + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [2 mat][2 parity][128][S]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t i0 = (int64_t)blockIdx.y * 128;
    const int j0 = blockIdx.x * 128;
    const bool vec_ok = (d % 4 == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    for (int k0 = 0; k0 < d; k0 += KC) {
        for (int f = t; f < 128 * (KC / 4); f += 256) {
            const int row = f / (KC / 4), v = f % (KC / 4);
            const int kk = k0 + 4 * v;
#pragma unroll
            for (int mat = 0; mat < 2; mat++) {
                const float* base = mat == 0 ? Q : Cn;
                const int64_t grow = mat == 0 ? i0 + row : (int64_t)j0 + row;
                const int64_t lim = mat == 0 ? nq : (int64_t)nlist;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (grow < lim) {
                    const float* p = base + grow * d + kk;
                    if (vec_ok && kk + 4 <= d) {
                        x = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (kk + 0 < d) x.x = p[0];
                        if (kk + 1 < d) x.y = p[1];
                        if (kk + 2 < d) x.z = p[2];
                        if (kk + 3 < d) x.w = p[3];
                    }
                }
                float* even = sm + ((mat * 2 + 0) * 128 + row) * S + 2 * v;
                float* odd = sm + ((mat * 2 + 1) * 128 + row) * S + 2 * v;
                *reinterpret_cast<float2*>(even) = make_float2(x.x, x.z);
                *reinterpret_cast<float2*>(odd) = make_float2(x.y, x.w);
            }
        }
        __syncthreads();
        const int h = lane >> 5, r = lane & 31;
#pragma unroll
        for (int u = 0; u < KC / 8; u++) {
            float4 a[2], b[2];
#pragma unroll
            for (int ti = 0; ti < 2; ti++) {
                a[ti] = *reinterpret_cast<const float4*>(
                    sm + ((0 * 2 + h) * 128 + wr * 64 + ti * 32 + r) * S + 4 * u);
                b[ti] = *reinterpret_cast<const float4*>(
                    sm + ((1 * 2 + h) * 128 + wc * 64 + ti * 32 + r) * S + 4 * u);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int ti = 0; ti < 2; ti++)
#pragma unroll
                    for (int tj = 0; tj < 2; tj++) {
                        const float av = e == 0 ? a[ti].x : e == 1 ? a[ti].y : e == 2 ? a[ti].z : a[ti].w;
                        const float bv = e == 0 ? b[tj].x : e == 1 ? b[tj].y : e == 2 ? b[tj].z : b[tj].w;
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ti][tj], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }

    // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int h = lane >> 5, cidx = lane & 31;
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++) {
            const int col = j0 + wc * 64 + tj * 32 + cidx;
            const float cnv = col < nlist ? cn[col] : 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int64_t row = i0 + wr * 64 + ti * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (row < nq && col < nlist) {
                    const float ip = acc[ti][tj][reg];
                    // (x_norm + y_norm) - 2*ip, utils.cpp:884
                    out[row * nlist + col] =
                        __fsub_rn(__fadd_rn(qn[row], cnv), __fmul_rn(2.f, ip));
                }
            }
        }
}

// ---------------------------------------------------------------------------
// coarse distances, d <= 128 (the BASELINE shapes): query fragments in registers.
// A workgroup owns 128 query rows (32 per wave) and walks `tiles_per_block` column
// tiles of 64 centroids.  The wave's A operands for the WHOLE k range (<= 64 k-pairs)
// are read once into 64 VGPRs, so LDS only holds the centroid tiles, double-buffered
// (2 x 34 KB -> two workgroups per CU): tile j+1 is fetched into registers while tile
// j is multiplied and written to the other buffer afterwards; one barrier per tile.
// Same k order (0,1,2,...) and epilogue as coarse_dist_kernel -> identical bits.
// ---------------------------------------------------------------------------
// TMIN == 1 (nlist % 64 == 0): also writes tmin[row][tile] = the minimum of the row's 64 distances in
// that column tile -- 1/64 of the matrix -- from which coarse_select_tiled_kernel finds the few
// tiles that can hold one of the nprobe nearest centroids and reads only those.
// TMIN == 3 (1-NN: the assignment of add / encode): NO matrix; per (row, tile) one 64-bit key
// ordered(distance) << 32 | column of the tile's (distance, column) minimum -- the total order of
// wave_topk.cuh -- so the nearest centroid of a row is the minimum of its nlist / 64 keys.
// TMIN == 5 (round 2, the filtered coarse stage): NO matrix either.  Every row comes with an upper bound
// of its nprobe-th smallest distance (the nprobe-th smallest over a SAMPLE of the columns, computed
// exactly by a first, small pass); an element at or below the bound is appended to the row's candidate
// list as a (distance, column) key, everything else is dropped.  The candidates (a few per cent of the
// row) are a superset of the row's nprobe nearest, so the exact select over them returns what the select
// over the full row returns.
constexpr int kFilterSlots = 16;     // kept keys per (row, 64-column tile); more -> the row is redone exactly
struct CoarseFilter {
    const float* bound;          // bound of row i = bound[i * stride]
    int64_t stride;
    unsigned long long* cand;    // [nq][ntiles][kFilterSlots] keys
    unsigned char* cnt;          // [nq][ntiles] keys kept in the tile (255: overflow)
    int ntiles;
};

template <int NU, bool VEC, int TMIN>   // k range padded to 8*NU; VEC: d % 4 == 0 (16-byte row loads)
__global__ __launch_bounds__(256, 2) void coarse_dist_areg_kernel(
    const float* __restrict__ Q, const float* __restrict__ Cn, const float* __restrict__ qn,
    const float* __restrict__ cn, float* __restrict__ out, int64_t nq, int nlist, int d,
    int tiles_per_block, float* __restrict__ tmin, CoarseFilter flt) {
    constexpr int KS = 4 * NU;        // k-pair steps
    constexpr int S = KS + 4;         // padded row stride (floats) of a parity plane
    extern __shared__ __attribute__((aligned(16))) float sm[];   // 2 buffers x [2 parity][64][S]
    constexpr int BUF = 2 * 64 * S;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int h = lane >> 5, r = lane & 31;
    float* tms = sm + 2 * BUF + wave * (32 * 17);     // TMIN: this wave's [32 rows][16 tiles + 1] staging
    const int64_t i0 = (int64_t)blockIdx.x * 128;
    const int tile0 = blockIdx.y * tiles_per_block;
    const int ntiles = (nlist + 63) / 64;

    // Branch-free loads: a conditional around a load makes hipcc wait for every element
    // separately (serialised L2 round trips).  Rows past the end are clamped -- their
    // products land in outputs that are never stored -- and the k padding is zeroed by a
    // select after the load.
    auto load4 = [&](const float* base, int64_t grow, int64_t lim, int kk) {
        const int64_t rowc = grow < lim ? grow : lim - 1;
        float4 x;
        if (VEC) {
            const int kc = kk < d ? kk : d - 4;
            x = *reinterpret_cast<const float4*>(base + rowc * d + kc);
            if (kk >= d) x = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const float* p = base + rowc * d;
            const int dm = d - 1;
            x.x = p[min(kk + 0, dm)];
            x.y = p[min(kk + 1, dm)];
            x.z = p[min(kk + 2, dm)];
            x.w = p[min(kk + 3, dm)];
            if (kk + 0 >= d) x.x = 0.f;
            if (kk + 1 >= d) x.y = 0.f;
            if (kk + 2 >= d) x.z = 0.f;
            if (kk + 3 >= d) x.w = 0.f;
        }
        return x;
    };

    float4 areg[NU];
    float* qns = sm + 2 * BUF + ((TMIN == 1 || TMIN == 8) ? 4 * 32 * 17 : 0);       // [128]
    if (VEC) {
        // ---- A fragments straight from global memory (round 3): lane (h, r) owns row wave*32 + r and, of every 8
        // consecutive k, the four of parity h -- two 16-byte loads per 8 k, no LDS staging, no workgroup barrier.
        // The row norm (qn == nullptr) falls out of the same loads in fvec_norm_L2sqr's order (utils.cpp:538-556:
        // lane sums over k = 4j + l, then (s0 + s1) + (s2 + s3); the zero-padded k add +0 to non-negative sums).
        const int64_t grow = i0 + wave * 32 + r;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const float4 x0 = load4(Q, grow, nq, 8 * u), x1 = load4(Q, grow, nq, 8 * u + 4);
            areg[u] = h ? make_float4(x0.y, x0.w, x1.y, x1.w) : make_float4(x0.x, x0.z, x1.x, x1.z);
            s0 = __fadd_rn(s0, __fmul_rn(x0.x, x0.x)); s1 = __fadd_rn(s1, __fmul_rn(x0.y, x0.y));
            s2 = __fadd_rn(s2, __fmul_rn(x0.z, x0.z)); s3 = __fadd_rn(s3, __fmul_rn(x0.w, x0.w));
            s0 = __fadd_rn(s0, __fmul_rn(x1.x, x1.x)); s1 = __fadd_rn(s1, __fmul_rn(x1.y, x1.y));
            s2 = __fadd_rn(s2, __fmul_rn(x1.z, x1.z)); s3 = __fadd_rn(s3, __fmul_rn(x1.w, x1.w));
        }
        if (!qn) {
            if (h == 0) qns[wave * 32 + r] = __fadd_rn(__fadd_rn(s0, s1), __fadd_rn(s2, s3));
            __builtin_amdgcn_wave_barrier();      // the 32 rows a wave reads below are the 32 it wrote
        }
    } else {
    // ---- stage the 128-row query tile through LDS (uses both buffers), pull A fragments
    for (int f = t; f < 128 * (2 * NU); f += 256) {
        const int row = f / (2 * NU), v = f % (2 * NU);
        const float4 x = load4(Q, i0 + row, nq, 4 * v);
        // rows 0..63 -> buffer 0, rows 64..127 -> buffer 1
        float* base = sm + (row >> 6) * BUF + (row & 63) * S + 2 * v;
        *reinterpret_cast<float2*>(base) = make_float2(x.x, x.z);
        *reinterpret_cast<float2*>(base + 64 * S) = make_float2(x.y, x.w);
    }
    __syncthreads();
    {
        const int row = wave * 32 + r;
        const float* src = sm + (row >> 6) * BUF + h * 64 * S + (row & 63) * S;
#pragma unroll
        for (int u = 0; u < NU; u++) areg[u] = *reinterpret_cast<const float4*>(src + 4 * u);
    }
    // qn == nullptr: the query norms are computed here, from the staged tile, in the reference's own order
    // (fvec_norms_L2sqr -> fvec_norm_L2sqr, utils.cpp:538-556, :675-682: norm_sse_order) -- one launch less
    // per coarse call; element k of a staged row sits in parity plane k & 1 at column k >> 1
    if (!qn) {
        if (t < 128) {
            const float* rowp = sm + (t >> 6) * BUF + (t & 63) * S;
            qns[t] = norm_sse_order([&](int c) { return rowp[(c & 1) * 64 * S + (c >> 1)]; }, d);
        }
        __syncthreads();
    }
    }
    float qnr[16];
    float bnd[TMIN == 5 ? 16 : 1];
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int lr = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int64_t row = i0 + lr;
        qnr[reg] = qn ? qn[row < nq ? row : nq - 1] : qns[lr];
        if (TMIN == 5) bnd[reg] = flt.bound[(row < nq ? row : nq - 1) * flt.stride];
    }
    __syncthreads();

    // ---- centroid tiles
    constexpr int NF = (64 * 2 * NU) / 256;     // float4 per thread per tile (= NU / 2)
    float4 pre[NF];
    auto fetch = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NF; i++) {
            const int f = t + 256 * i;
            const int row = f / (2 * NU), v = f % (2 * NU);
            pre[i] = load4(Cn, (int64_t)tile * 64 + row, nlist, 4 * v);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NF; i++) {
            const int f = t + 256 * i;
            const int row = f / (2 * NU), v = f % (2 * NU);
            float* base = sm + buf * BUF + row * S + 2 * v;
            *reinterpret_cast<float2*>(base) = make_float2(pre[i].x, pre[i].z);
            *reinterpret_cast<float2*>(base + 64 * S) = make_float2(pre[i].y, pre[i].w);
        }
    };
    const int tend = min(ntiles, tile0 + tiles_per_block);
    if (tile0 >= tend) return;
    fetch(tile0);
    stash(0);
    if (tile0 + 1 < tend) fetch(tile0 + 1);
    __syncthreads();
    if (TMIN == 7 || TMIN == 8) {     // 8 = 7 + the per-tile row minima of TMIN == 1
        // Software pipeline over HALF tiles (round 3): the matrix stores of a tile were what the plain loop paid 33
        // of its 125 us for -- the epilogue of a tile sat between its MFMAs and the next tile's.  Here the two
        // 32-column halves of a tile keep their two accumulators but are multiplied one after the other, and the
        // 64 MFMAs of one half and the epilogue of the half before it -- distances, 4x4 transposes, 16-byte stores --
        // are ONE basic block (every row of the workgroup and every column of the tile exists: the launcher hands
        // partial row blocks to the plain kernel), scheduled so that the VALU work and the stores issue in the MFMA
        // shadow.  No register more than the plain loop: two workgroups per CU as before.  Same products, same k
        // order per element, same epilogue arithmetic: bit-identical output.
        float pmrow[4] = {0.f, 0.f, 0.f, 0.f};     // TMIN == 8: running minimum of the tile's columns per row group
        // The epilogue of one 4-register group (4 consecutive rows x this lane's column) in four parts, so that it can
        // be dealt out between the MFMAs of the next half: 0 / 1 distances of registers 0,1 / 2,3 ((x_norm + y_norm)
        // - 2*ip, utils.cpp:884), 2 / 3 the two exchange steps of the 4x4 transpose inside a lane quad; part 3 ends
        // with the 16-byte store (one row x 4 consecutive columns).
        auto epi_part = [&](int m, int g, int tile, int tj, const f32x16& acc, float cnv, float (&v)[4]) __attribute__((always_inline)) {
            if (m < 2) {
#pragma unroll
                for (int i = 2 * m; i < 2 * m + 2; i++)
                    v[i] = __fsub_rn(__fadd_rn(qnr[4 * g + i], cnv), __fmul_rn(2.f, acc[4 * g + i]));
            } else if (m == 2) {
                const bool odd = lane & 1;
                float s0 = odd ? v[0] : v[1], s1 = odd ? v[2] : v[3];
                s0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0xB1, 0xf, 0xf, false));
                s1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1, 0xf, 0xf, false));
                if (odd) { v[0] = s0; v[2] = s1; } else { v[1] = s0; v[3] = s1; }
            } else {
                const bool up = lane & 2;
                float s0 = up ? v[0] : v[2], s1 = up ? v[1] : v[3];
                s0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0x4E, 0xf, 0xf, false));
                s1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x4E, 0xf, 0xf, false));
                if (up) { v[0] = s0; v[1] = s1; } else { v[2] = s0; v[3] = s1; }
                const int64_t row = i0 + wave * 32 + 8 * g + 4 * h + (r & 3);
                *reinterpret_cast<float4*>(out + row * nlist + tile * 64 + tj * 32 + (r & ~3)) = make_float4(v[0], v[1], v[2], v[3]);
                if (TMIN == 8) {
                    const float m4 = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
                    pmrow[g] = tj == 0 ? m4 : fminf(pmrow[g], m4);
                }
            }
        };
        auto epilogue_half = [&](int tile, int tj, const f32x16& acc, float cnv) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float v[4];
#pragma unroll
                for (int m = 0; m < 4; m++) epi_part(m, g, tile, tj, acc, cnv, v);
            }
        };
        // One half tile: the 4 NU MFMAs of (buf, tj) into acc and -- EPI -- the epilogue of the half before it
        // (ptile, ptj, pacc), dealt out behind the MFMAs: a group of 4 dependent MFMAs occupies the pipe for 256 clocks,
        // and the at most 12 VALU operations + one store issued after it finish inside the last one's 64.  The
        // scheduler clustered all MFMAs in front of the whole epilogue whatever sched_group_barrier pattern asked for, so
        // the order is pinned with full scheduling barriers here; the B operand of the next k group is read one group
        // ahead.  Round-4 decomposition at C1 (10 000 x 4096 x 128, 474 workgroups on 512 slots, clocks 2.08 GHz under
        // this load: the busiest SIMD's MFMAs alone are 87 us): loop without epilogue, staging and barrier 96 us;
        // + epilogue arithmetic 14 (pinned or not), + matrix stores 15 (non-temporal: same), + staging 7.5,
        // + barrier 1.6 = 127 us.
        auto half_step = [&](auto epi, int buf, int tj, f32x16& acc, int ptile, int ptj, const f32x16& pacc, float pcn) __attribute__((always_inline)) {
            constexpr bool EPI = decltype(epi)::value;
            constexpr int P = NU / 4;                 // k groups per register group of the epilogue
            const float* bsrc = sm + buf * BUF + h * 64 * S + (tj * 32 + r) * S;
            float4 bcur = *reinterpret_cast<const float4*>(bsrc);
            float v[4];
#pragma unroll
            for (int reg = 0; reg < 16; reg++) acc[reg] = 0.f;
#pragma unroll
            for (int u = 0; u < NU; u++) {
                float4 bnext = bcur;
                if (u + 1 < NU) bnext = *reinterpret_cast<const float4*>(bsrc + 4 * (u + 1));
                const float4 av = areg[u];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bcur.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bcur.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bcur.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bcur.w, acc, 0, 0, 0);
                if (EPI) {
#pragma unroll
                    for (int m = 0; m < 4; m++)
                        if (m * P / 4 == u % P) epi_part(m, u / P, ptile, ptj, pacc, pcn, v);
                }
                bcur = bnext;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // TMIN == 8, after both halves of a tile: the rows' tile minima (over lane bits 2..4), staged per wave and
        // written 16 tiles at a time -- the code of the plain loop's TMIN == 1 branch
        auto tile_minima = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float m = pmrow[g];
                m = fminf(m, __uint_as_float(lane_xor_u32(__float_as_uint(m), 4)));
                m = fminf(m, __uint_as_float(lane_xor_u32(__float_as_uint(m), 8)));
                m = fminf(m, __uint_as_float(lane_xor_u32(__float_as_uint(m), 16)));
                if (r < 4) tms[(8 * g + 4 * h + r) * 17 + ((tile - tile0) & 15)] = m;
            }
            const int done = tile - tile0 + 1;
            if ((done & 15) == 0 || tile + 1 == tend) {
                __builtin_amdgcn_wave_barrier();
                const int ncol = ((done - 1) & 15) + 1, t0 = tile + 1 - ncol, c = lane & 15;
#pragma unroll
                for (int rr = 0; rr < 32; rr += 4) {
                    const int lr = rr + (lane >> 4);
                    const int64_t row = i0 + wave * 32 + lr;
                    if (c < ncol && row < nq) tmin[row * (nlist >> 6) + t0 + c] = tms[lr * 17 + c];
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        f32x16 acc0, acc1;
        const std::true_type with_epilogue{};
        float cn0 = cn[tile0 * 64 + r], cn1 = 0.f;    // column norms of the halves in flight, requested one half ahead
        half_step(std::false_type{}, 0, 0, acc0, 0, 0, acc0, 0.f);      // first tile, first half: nothing to overlap with yet
        for (int tile = tile0; tile < tend; tile++) {
            const int buf = (tile - tile0) & 1;
            cn1 = cn[tile * 64 + 32 + r];
            half_step(with_epilogue, buf, 1, acc1, tile, 0, acc0, cn0);
            // every wave has read its B operands of this tile: the next tile's registers go to the other buffer
            // (free since the barrier of the previous iteration), the tile after it is requested
            if (tile + 1 < tend) {
                stash(buf ^ 1);
                if (tile + 2 < tend) fetch(tile + 2);
            }
            __syncthreads();
            if (tile + 1 < tend) {
                cn0 = cn[(tile + 1) * 64 + r];
                half_step(with_epilogue, buf ^ 1, 0, acc0, tile, 1, acc1, cn1);
            } else {
                epilogue_half(tile, 1, acc1, cn1);
            }
            if (TMIN == 8) tile_minima(tile);
        }
        return;
    }
    for (int tile = tile0; tile < tend; tile++) {
        const int buf = (tile - tile0) & 1;
        f32x16 acc[2];
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++) acc[tj][reg] = 0.f;
        const float* bsrc = sm + buf * BUF + h * 64 * S + r * S;
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const float4 b0 = *reinterpret_cast<const float4*>(bsrc + 4 * u);
            const float4 b1 = *reinterpret_cast<const float4*>(bsrc + 32 * S + 4 * u);
            const float4 av = areg[u];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc[1], 0, 0, 0);
        }
        // next tile: registers -> the other buffer (free since the barrier of the previous
        // iteration), then request the tile after it
        if (tile + 1 < tend) {
            stash(buf ^ 1);
            if (tile + 2 < tend) fetch(tile + 2);
        }
        // epilogue of this tile.  A lane holds one column and, per group of 4 registers,
        // 4 consecutive rows; a 4x4 transpose inside each lane quad (two DPP exchange
        // steps) turns that into one row x 4 consecutive columns, so the tile leaves as
        // 16-byte stores that cover whole 128-byte lines (8 rows per instruction) --
        // 4x fewer store instructions than dword stores, which were issue-bound.
        const bool full = (tile * 64 + 64 <= nlist) && ((nlist & 3) == 0);
        float mrow[4];
        u64 krow[4];
        if (TMIN == 5) {
            // lane (r, h) holds columns tile*64 + tj*32 + r of rows 8g + 4h + i (reg = 4g + i): for every row the
            // 32 lanes of a half-wave hold 32 of its columns.  Elements at or below the row's bound are compacted
            // into the (row, tile) slot group with a ballot -- no atomics, no matrix.
            const uint32_t lt = (1u << r) - 1u;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int64_t row = i0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                int base = 0;
#pragma unroll
                for (int tj = 0; tj < 2; tj++) {
                    const int col = tile * 64 + tj * 32 + r;
                    const float cnv = cn[col < nlist ? col : nlist - 1];
                    const float x = __fsub_rn(__fadd_rn(qnr[reg], cnv), __fmul_rn(2.f, acc[tj][reg]));
                    const bool pass = row < nq && col < nlist && x <= bnd[TMIN == 5 ? reg : 0] && x < FLT_MAX_F;
                    const u64 m64 = __ballot(pass);
                    const uint32_t mh = h ? (uint32_t)(m64 >> 32) : (uint32_t)m64;
                    const int sl = base + __popc(mh & lt);
                    if (pass && sl < kFilterSlots)
                        flt.cand[((size_t)row * flt.ntiles + tile) * kFilterSlots + sl] = make_key(x, (uint32_t)col);
                    base += __popc(mh);
                }
                if (r == 0 && row < nq) flt.cnt[(size_t)row * flt.ntiles + tile] = (unsigned char)(base > kFilterSlots ? 255 : base);
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int tj = 0; tj < 2; tj++) {
            const int col = tile * 64 + tj * 32 + r;
            const float cnv = cn[col < nlist ? col : nlist - 1];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++)   // (x_norm + y_norm) - 2*ip, utils.cpp:884
                    v[i] = __fsub_rn(__fadd_rn(qnr[4 * g + i], cnv), __fmul_rn(2.f, acc[tj][4 * g + i]));
                if (full) {
                    // step 1: lanes l, l^1 exchange across register pairs (0,1) (2,3)
                    {
                        const bool odd = lane & 1;
                        float s0 = odd ? v[0] : v[1], s1 = odd ? v[2] : v[3];
                        s0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0xB1, 0xf, 0xf, false));
                        s1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1, 0xf, 0xf, false));
                        if (odd) { v[0] = s0; v[2] = s1; } else { v[1] = s0; v[3] = s1; }
                    }
                    // step 2: lanes l, l^2 exchange across register pairs (0,2) (1,3)
                    {
                        const bool up = lane & 2;
                        float s0 = up ? v[0] : v[2], s1 = up ? v[1] : v[3];
                        s0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0x4E, 0xf, 0xf, false));
                        s1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x4E, 0xf, 0xf, false));
                        if (up) { v[0] = s0; v[1] = s1; } else { v[2] = s0; v[3] = s1; }
                    }
                    // lane (r = 4q + i, h): row 8g + 4h + i, columns 4q .. 4q+3
                    const int64_t row = i0 + wave * 32 + 8 * g + 4 * h + (r & 3);
                    if (TMIN != 3 && row < nq)
                        *reinterpret_cast<float4*>(out + row * nlist + tile * 64 + tj * 32 + (r & ~3)) =
                            make_float4(v[0], v[1], v[2], v[3]);
                    if (TMIN == 1) {
                        const float m4 = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
                        mrow[g] = tj == 0 ? m4 : fminf(mrow[g], m4);
                    }
                    if (TMIN == 3) {
                        const uint32_t c0 = (uint32_t)(tile * 64 + tj * 32 + (r & ~3));
                        u64 kk = make_key(v[0], c0);
                        kk = umin64(kk, make_key(v[1], c0 + 1));
                        kk = umin64(kk, make_key(v[2], c0 + 2));
                        kk = umin64(kk, make_key(v[3], c0 + 3));
                        krow[g] = tj == 0 ? kk : umin64(krow[g], kk);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int64_t row = i0 + wave * 32 + 8 * g + 4 * h + i;
                        if (row < nq && col < nlist) out[row * nlist + col] = v[i];
                    }
                }
            }
        }
        if (TMIN == 3) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                u64 kk = krow[g];
                kk = umin64(kk, shfl_xor_u64(kk, 4));
                kk = umin64(kk, shfl_xor_u64(kk, 8));
                kk = umin64(kk, shfl_xor_u64(kk, 16));
                const int64_t row = i0 + wave * 32 + 8 * g + 4 * h + r;
                if (r < 4 && row < nq) reinterpret_cast<u64*>(tmin)[row * (nlist >> 6) + tile] = kk;
            }
        }
        if (TMIN == 1) {
            // lane (r = 4q + i, h) holds row 8g + 4h + i, columns 4q..4q+3 of both halves: the row's
            // tile minimum is the minimum over q, i.e. over lane bits 2..4
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float m = mrow[g];
                m = fminf(m, __uint_as_float(lane_xor_u32(__float_as_uint(m), 4)));
                m = fminf(m, __uint_as_float(lane_xor_u32(__float_as_uint(m), 8)));
                m = fminf(m, __uint_as_float(lane_xor_u32(__float_as_uint(m), 16)));
                if (r < 4) tms[(8 * g + 4 * h + r) * 17 + ((tile - tile0) & 15)] = m;
            }
            // a wave owns its 32 rows: every 16 tiles (and at the end) it writes 16 consecutive
            // minima per row, 64 contiguous bytes, instead of 4-byte stores one matrix row apart
            const int done = tile - tile0 + 1;
            if ((done & 15) == 0 || tile + 1 == tend) {
                __builtin_amdgcn_wave_barrier();
                const int ncol = ((done - 1) & 15) + 1, t0 = tile + 1 - ncol, c = lane & 15;
#pragma unroll
                for (int rr = 0; rr < 32; rr += 4) {
                    const int lr = rr + (lane >> 4);
                    const int64_t row = i0 + wave * 32 + lr;
                    if (c < ncol && row < nq) tmin[row * (nlist >> 6) + t0 + c] = tms[lr * 17 + c];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
    }
}

template <int NU, bool VEC, int TMIN>
static void launch_coarse_areg_t(const float* q, const float* c, const float* qn, const float* cn,
                               float* out, int64_t nq, int nlist, int d, float* tmin, hipStream_t s,
                               CoarseFilter flt = CoarseFilter{nullptr, 0, nullptr, nullptr, 0}) {
    constexpr int S = 4 * NU + 4;
    const size_t smem = (size_t)2 * 2 * 64 * S * sizeof(float) + ((TMIN == 1 || TMIN == 8) ? (size_t)4 * 32 * 17 * sizeof(float) : 0) +
                        128 * sizeof(float);      // + the fused query norms
    ensure_dynamic_lds(reinterpret_cast<const void*>(coarse_dist_areg_kernel<NU, VEC, TMIN>), smem);
    const int64_t rb = (nq + 127) / 128;
    const int ntiles = (nlist + 63) / 64;
    // tiles per workgroup: the MFMA pipe of a CU is shared by its (up to 2) resident
    // workgroups, so the time is ~ (tiles + prologue) of the busiest CU over all rounds.  The prologue (query
    // tile through LDS, norms, first centroid tile) costs about 1.2 tile times (round 3, measured at C1 under
    // rocprof: 4 tiles per workgroup 128 us, 11 tiles 121 us, 22 tiles -- one workgroup per CU, nothing to overlap
    // with -- 132 us); a CU with a single resident workgroup loses the overlap of the second one.
    int best_t = 1;
    double best_cost = 1e30;
    for (int tpb = 1; tpb <= 32; tpb++) {
        const int64_t blocks = rb * ((ntiles + tpb - 1) / tpb);
        const int64_t per_cu = (blocks + 255) / 256;            // workgroups the busiest CU runs
        double cost = (double)per_cu * (tpb + 1.2);
        if (per_cu < 2 && blocks > 256 / 2) cost *= 1.1;        // lone workgroups: nothing hides their barriers
        if (cost < best_cost) { best_cost = cost; best_t = tpb; }
    }
    if (const char* e = getenv("VLQ_COARSE_TPB")) { const int v = atoi(e); if (v >= 1 && v <= 64) best_t = v; }   // A/B only
    dim3 grid((unsigned)rb, (unsigned)((ntiles + best_t - 1) / best_t));
    hipLaunchKernelGGL((coarse_dist_areg_kernel<NU, VEC, TMIN>), grid, dim3(256), smem, s, q, c, qn, cn, out,
                       nq, nlist, d, best_t, tmin, flt);
}

template <int NU>
static void launch_coarse_areg(const float* q, const float* c, const float* qn, const float* cn,
                               float* out, int64_t nq, int nlist, int d, float* tmin, hipStream_t s, int64_t out_rows) {
    if (tmin && !out) launch_coarse_areg_t<NU, true, 3>(q, c, qn, cn, out, nq, nlist, d, tmin, s);   // coarse_argmin_ok
    else if (tmin && out_rows >= (nq + 127) / 128 * 128 && !getenv("VLQ_COARSE_PLAIN"))
        launch_coarse_areg_t<NU, true, 8>(q, c, qn, cn, out, nq, nlist, d, tmin, s);                   // coarse_tile_minima_ok, pipelined
    else if (tmin) launch_coarse_areg_t<NU, true, 1>(q, c, qn, cn, out, nq, nlist, d, tmin, s);       // coarse_tile_minima_ok
    else if (d % 4 == 0 && d >= 4 && nlist % 64 == 0 && out_rows >= (nq + 127) / 128 * 128 && !getenv("VLQ_COARSE_PLAIN")) {
        // plain matrix, whole tiles, and the matrix has room for whole 128-row blocks (rows past nq are written and
        // never read): the pipelined loop, whose epilogue has no row guard
        launch_coarse_areg_t<NU, true, 7>(q, c, qn, cn, out, nq, nlist, d, nullptr, s);
    }
    else if (d % 4 == 0 && d >= 4) launch_coarse_areg_t<NU, true, 0>(q, c, qn, cn, out, nq, nlist, d, nullptr, s);
    else launch_coarse_areg_t<NU, false, 0>(q, c, qn, cn, out, nq, nlist, d, nullptr, s);
}

// tile minima are produced by the d <= 128 kernels for whole 64-column tiles and 16-byte row loads
bool coarse_tile_minima_ok(int nlist, int d, int nprobe) {
    return d <= 128 && d >= 4 && d % 4 == 0 && nlist % 64 == 0 && nlist > 8192 && nlist <= 2048 * 64 &&
           nlist / 64 >= 4 * nprobe;
}

// 1-NN without a distance matrix (out == nullptr, tmin = [nq][nlist / 64] 64-bit keys)
bool coarse_argmin_ok(int nlist, int d) {
    return d <= 128 && d >= 4 && d % 4 == 0 && nlist % 64 == 0 && nlist >= 256;
}

// ---------------------------------------------------------------------------
// filtered coarse stage (round 2): sample pass -> per-row bound -> full pass that keeps only the
// elements at or below the bound -> exact select over the kept keys.  Exact: the bound is the
// nprobe-th smallest distance over a subset of the row's own elements, so at least nprobe elements
// lie at or below it and every one of the row's nprobe nearest is among the kept.  A row whose kept
// elements do not fit its candidate buffer (long runs of equal distances) is redone from scratch by
// the select kernel: the inner products as k-ascending fmaf chains, bit-identical to the MFMA
// accumulation.
// ---------------------------------------------------------------------------
bool coarse_filter_ok(int nlist, int d, int nprobe, int64_t nq, int* stride_out, int* cap_out) {
    if (!(d <= 128 && d >= 4 && d % 4 == 0 && nlist % 64 == 0 && nprobe >= 2 && nprobe <= 256 && nq >= 256)) return false;
    int s = 1;
    while (s * 2 <= 16 && (int64_t)nlist / (s * 2) >= (int64_t)16 * nprobe && (nlist / 64) % (s * 2) == 0) s *= 2;
    if (s < 4) return false;
    *stride_out = s;
    *cap_out = kFilterSlots;
    return true;
}

// the sampled columns: tiles 0, s, 2s, ... of the centroid matrix, packed
__global__ void sample_tiles_kernel(const float* __restrict__ c, const float* __restrict__ cn, int nlist, int d, int s,
                                    float* __restrict__ cs, float* __restrict__ cns) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ns = nlist / s;
    if (e >= (int64_t)ns * d) return;
    const int64_t j = e / d;                 // sampled column
    const int k = (int)(e % d);
    const int64_t src = (j / 64) * 64 * s + (j % 64);
    cs[e] = c[src * d + k];
    if (k == 0) cns[j] = cn[src];
}
void launch_sample_tiles(const float* c, const float* cn, int nlist, int d, int s, float* cs, float* cns, hipStream_t st) {
    const int64_t tot = (int64_t)(nlist / s) * d;
    hipLaunchKernelGGL(sample_tiles_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, c, cn, nlist, d, s, cs, cns);
}

void launch_coarse_distances_filtered(const float* q, const float* c, const float* qn, const float* cn, int64_t nq, int nlist,
                                      int d, const float* bound, int64_t bound_stride, unsigned long long* cand,
                                      unsigned char* cnt, hipStream_t s) {
    CoarseFilter f{bound, bound_stride, cand, cnt, nlist / 64};
    if (d <= 32) launch_coarse_areg_t<4, true, 5>(q, c, qn, cn, nullptr, nq, nlist, d, nullptr, s, f);
    else if (d <= 64) launch_coarse_areg_t<8, true, 5>(q, c, qn, cn, nullptr, nq, nlist, d, nullptr, s, f);
    else if (d <= 96) launch_coarse_areg_t<12, true, 5>(q, c, qn, cn, nullptr, nq, nlist, d, nullptr, s, f);
    else launch_coarse_areg_t<16, true, 5>(q, c, qn, cn, nullptr, nq, nlist, d, nullptr, s, f);
}

// exact select over a row's kept keys; a row with an overflowed tile is recomputed in full
template <int KPL>
__global__ __launch_bounds__(256) void coarse_select_cand_kernel(const u64* __restrict__ cand, const unsigned char* __restrict__ cnt,
                                                                 int ntiles, int64_t nq, int nprobe, float* __restrict__ cdis,
                                                                 int64_t* __restrict__ keys, const float* __restrict__ Q,
                                                                 const float* __restrict__ Cn, const float* __restrict__ qn,
                                                                 const float* __restrict__ cn, int nlist, int d) {
    __shared__ u64 queue[4][64];
    __shared__ float qrow[4][128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;   // whole wave; no workgroup barrier below
    WaveSelect<KPL> sel;
    sel.init(nprobe, queue[wave], lane);
    const unsigned char* crow = cnt + q * ntiles;
    bool ovf = false;
    for (int t0 = 0; t0 < ntiles; t0 += 64) {            // a lane per tile
        const int tl = t0 + lane;
        const int c = tl < ntiles ? (int)crow[tl] : 0;
        ovf = ovf || __ballot(c == 255) != 0;
    }
    if (!ovf) {
        for (int t0 = 0; t0 < ntiles; t0 += 64) {
            const int tl = t0 + lane;
            const int c = tl < ntiles ? (int)crow[tl] : 0;
            int cmax = c;
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) cmax = max(cmax, __shfl_xor(cmax, sft, 64));
            const u64* grp = cand + ((size_t)q * ntiles + (tl < ntiles ? tl : 0)) * kFilterSlots;
            for (int j = 0; j < cmax; j++) {
                const bool valid = j < c;
                const u64 key = valid ? grp[j] : kMaxKey;
                // columns do not arrive in increasing order: equal distances are queued, the key decides
                sel.template offer<false>(ordered_to_f32((uint32_t)(key >> 32)), (uint32_t)key, valid);
            }
        }
    } else {
        // the whole row again, one column per lane: ip = fmaf chain over k = 0, 1, 2, ... (what the MFMA
        // accumulates), value = (|q|^2 + |c|^2) - 2 ip (utils.cpp:884)
        for (int k = lane; k < d; k += 64) qrow[wave][k] = Q[q * d + k];
        __builtin_amdgcn_wave_barrier();
        const float qnv = qn[q];
        for (int j0 = 0; j0 < nlist; j0 += 64) {
            const int j = j0 + lane;
            const bool valid = j < nlist;
            const float* cj = Cn + (size_t)(valid ? j : 0) * d;
            float ip = 0.f;
            for (int k = 0; k < d; k++) ip = __fmaf_rn(qrow[wave][k], cj[k], ip);
            const float v = __fsub_rn(__fadd_rn(qnv, cn[valid ? j : 0]), __fmul_rn(2.f, ip));
            sel.offer(v, (uint32_t)j, valid);          // ascending columns: the ordered rule is exact
        }
    }
    sel.flush();
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = r * 64 + lane;
        if (e < nprobe) {
            const u64 key = sel.best[r];
            const bool miss = key == kMaxKey;
            cdis[q * nprobe + e] = miss ? FLT_MAX_F : ordered_to_f32((uint32_t)(key >> 32));
            keys[q * nprobe + e] = miss ? -1 : (int64_t)(uint32_t)key;
        }
    }
}

void launch_coarse_select_cand(const unsigned long long* cand, const unsigned char* cnt, int64_t nq, int nprobe, float* cdis,
                               int64_t* keys, const float* q, const float* c, const float* qn, const float* cn, int nlist, int d,
                               hipStream_t s) {
    if (nq <= 0) return;
    dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    const u64* cd = reinterpret_cast<const u64*>(cand);
    const int ntiles = nlist / 64;
    if (nprobe <= 64) hipLaunchKernelGGL(coarse_select_cand_kernel<1>, grid, block, 0, s, cd, cnt, ntiles, nq, nprobe, cdis, keys, q, c, qn, cn, nlist, d);
    else hipLaunchKernelGGL(coarse_select_cand_kernel<4>, grid, block, 0, s, cd, cnt, ntiles, nq, nprobe, cdis, keys, q, c, qn, cn, nlist, d);
}

// nearest centroid of every row from the per-tile keys of the TMIN == 3 distance kernel
__global__ __launch_bounds__(256) void coarse_argmin_kernel(const u64* __restrict__ tkeys, int64_t nq, int ntiles,
                                                            float* __restrict__ cdis, int64_t* __restrict__ keys) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const u64* row = tkeys + q * ntiles;
    u64 best = kMaxKey;
    for (int j = lane; j < ntiles; j += 64) best = umin64(best, row[j]);
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) best = umin64(best, shfl_xor_u64(best, sft));
    if (lane == 0) {
        const float dis = ordered_to_f32((uint32_t)(best >> 32));
        // the reference's heap starts at FLT_MAX and admits only dis < top (Heap.h:76-78)
        const bool miss = !(dis < FLT_MAX_F);
        cdis[q] = miss ? FLT_MAX_F : dis;
        keys[q] = miss ? -1 : (int64_t)(uint32_t)best;
    }
}

void launch_coarse_argmin(const void* tkeys, int64_t nq, int nlist, float* cdis, int64_t* keys, hipStream_t s) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(coarse_argmin_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s,
                       reinterpret_cast<const u64*>(tkeys), nq, nlist >> 6, cdis, keys);
}

void launch_coarse_distances(const float* q, const float* c, const float* qn, const float* cn,
                             float* out, int64_t nq, int nlist, int d, hipStream_t s, float* tmin, int64_t out_rows) {
    if (nq <= 0 || nlist <= 0) return;
    if (d <= 128) {
        if (d <= 32) launch_coarse_areg<4>(q, c, qn, cn, out, nq, nlist, d, tmin, s, out_rows);
        else if (d <= 64) launch_coarse_areg<8>(q, c, qn, cn, out, nq, nlist, d, tmin, s, out_rows);
        else if (d <= 96) launch_coarse_areg<12>(q, c, qn, cn, out, nq, nlist, d, tmin, s, out_rows);
        else launch_coarse_areg<16>(q, c, qn, cn, out, nq, nlist, d, tmin, s, out_rows);
        return;
    }
    constexpr int KC = 64;
    const size_t smem = 2 * 2 * 128 * (KC / 2 + 4) * sizeof(float);
    ensure_dynamic_lds(reinterpret_cast<const void*>(coarse_dist_kernel<KC>), smem);
    dim3 grid((unsigned)((nlist + 127) / 128), (unsigned)((nq + 127) / 128));
    hipLaunchKernelGGL(coarse_dist_kernel<KC>, grid, dim3(256), smem, s, q, c, qn, cn, out, nq,
                       nlist, d);
}

// ---------------------------------------------------------------------------
// coarse distances for batches of < 20 queries: the reference then skips BLAS and
// calls fvec_L2sqr per (query, centroid) pair (knn_L2sqr_sse, utils.cpp:757-786,
// dispatch :935-946).  Same operation order here, one thread per centroid.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coarse_direct_kernel(const float* __restrict__ Q,
                                                            const float* __restrict__ Cn,
                                                            float* __restrict__ out, int64_t nq,
                                                            int nlist, int d) {
    extern __shared__ __attribute__((aligned(16))) float sq[];   // [d]
    const int64_t i = blockIdx.y;
    for (int c = threadIdx.x; c < d; c += 256) sq[c] = Q[i * d + c];
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nlist) return;
    const float* cj = Cn + (size_t)j * d;
    out[i * nlist + j] =
        l2sqr_sse_order([&](int c) { return sq[c]; }, [&](int c) { return cj[c]; }, d);
}

void launch_coarse_distances_direct(const float* q, const float* c, float* out, int64_t nq,
                                    int nlist, int d, hipStream_t s) {
    if (nq <= 0 || nlist <= 0) return;
    dim3 grid((unsigned)((nlist + 255) / 256), (unsigned)nq);
    hipLaunchKernelGGL(coarse_direct_kernel, grid, dim3(256), (size_t)d * sizeof(float), s, q, c,
                       out, nq, nlist, d);
}

// ---------------------------------------------------------------------------
// coarse select: one wave per query row, running selection of the nprobe
// smallest (distance, centroid id).
// ---------------------------------------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void coarse_select_kernel(const float* __restrict__ dist,
                                                            int64_t nq, int nlist, int nprobe,
                                                            float* __restrict__ cdis,
                                                            int64_t* __restrict__ keys) {
    __shared__ u64 queue[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;   // whole wave; no workgroup barrier below
    WaveSelect<KPL> sel;
    sel.init(nprobe, queue[wave], lane);
    const float* row = dist + q * nlist;
    if ((nlist & 3) == 0) {
        // 16 B per lane; a wave walks its row alone, so the loop is a chain of load latencies unless
        // several KiB are in flight: 4 x 1 KiB requested ahead of the one being consumed
        const float4* row4 = reinterpret_cast<const float4*>(row);
        const int n4 = nlist >> 2;
        constexpr int PF = 4;
        float4 buf[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) buf[u] = row4[min(u * 64 + lane, n4 - 1)];
        for (int j0 = 0; j0 < n4; j0 += 64 * PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int j4 = j0 + u * 64 + lane;
                const float4 cur = buf[u];
                buf[u] = row4[min(j4 + 64 * PF, n4 - 1)];          // clamped, unconditional
                const bool valid = j4 < n4;
                // columns 4*j4+c, c = 0..3: not in increasing order across the four offers
                sel.template offer<false>(cur.x, (uint32_t)(4 * j4 + 0), valid);
                sel.template offer<false>(cur.y, (uint32_t)(4 * j4 + 1), valid);
                sel.template offer<false>(cur.z, (uint32_t)(4 * j4 + 2), valid);
                sel.template offer<false>(cur.w, (uint32_t)(4 * j4 + 3), valid);
            }
        }
    } else {
        for (int j0 = 0; j0 < nlist; j0 += 64) {
            const int j = j0 + lane;
            const bool valid = j < nlist;
            const float v = valid ? row[j] : 0.f;
            sel.offer(v, (uint32_t)j, valid);
        }
    }
    sel.flush();
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = r * 64 + lane;
        if (e < nprobe) {
            const u64 key = sel.best[r];
            const bool miss = key == kMaxKey;
            cdis[q * nprobe + e] = miss ? FLT_MAX_F : ordered_to_f32((uint32_t)(key >> 32));
            keys[q * nprobe + e] = miss ? -1 : (int64_t)(uint32_t)key;
        }
    }
}

// Row-in-registers select for nprobe <= 64 and rows of at most 256 * NV floats (nlist = 4096:
// NV = 16): a wave pulls its whole row with NV outstanding 16-byte loads per lane, takes the
// nprobe-th smallest of the 64 per-lane minima as a cut -- an upper bound of the nprobe-th
// smallest element, because nprobe distinct elements lie at or below it -- and only the few
// dozen elements at or below the cut enter the exact (distance, column) selection.  Elements at
// FLT_MAX are never admitted (the reference heap starts there, Heap.h:76-78).
template <int NV>
__global__ __launch_bounds__(256) void coarse_select_reg_kernel(const float* __restrict__ dist, int64_t nq,
                                                                int nlist, int nprobe, float* __restrict__ cdis,
                                                                int64_t* __restrict__ keys) {
    constexpr int CAP = 512;                    // candidates at or below the cut, per row
    __shared__ u64 queue[4][64];
    __shared__ u64 cand[4][CAP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const float4* row4 = reinterpret_cast<const float4*>(dist + q * nlist);
    const int n4 = nlist >> 2;
    float4 v[NV];
#pragma unroll
    for (int u = 0; u < NV; u++) v[u] = row4[min(u * 64 + lane, n4 - 1)];    // clamped; masked below
    float mn = FLT_MAX_F;
#pragma unroll
    for (int u = 0; u < NV; u++) {
        if (u * 64 + lane >= n4) v[u] = make_float4(FLT_MAX_F, FLT_MAX_F, FLT_MAX_F, FLT_MAX_F);
        mn = fminf(mn, fminf(fminf(v[u].x, v[u].y), fminf(v[u].z, v[u].w)));
    }
    // cut = nprobe-th smallest lane minimum (lanes whose minimum is FLT_MAX hold nothing admissible)
    const u64 sorted = wave_sort64(((u64)f32_to_ordered(mn) << 32) | (uint32_t)lane, lane);
    const float cut = ordered_to_f32((uint32_t)(bcast_u64(sorted, nprobe - 1) >> 32));
    auto pass = [&](float x) { return x <= cut && x < FLT_MAX_F; };
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < NV; u++) cnt += (int)pass(v[u].x) + (int)pass(v[u].y) + (int)pass(v[u].z) + (int)pass(v[u].w);
    int incl = cnt;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const int o = __shfl_up(incl, sft, 64);
        if (lane >= sft) incl += o;
    }
    const int total = __shfl(incl, 63, 64);
    WaveSelect<1> sel;
    sel.init(nprobe, queue[wave], lane);
    if (total <= CAP) {
        int pos = incl - cnt;
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const uint32_t c0 = (uint32_t)(4 * (u * 64 + lane));
            if (pass(v[u].x)) cand[wave][pos++] = make_key(v[u].x, c0 + 0);
            if (pass(v[u].y)) cand[wave][pos++] = make_key(v[u].y, c0 + 1);
            if (pass(v[u].z)) cand[wave][pos++] = make_key(v[u].z, c0 + 2);
            if (pass(v[u].w)) cand[wave][pos++] = make_key(v[u].w, c0 + 3);
        }
        __builtin_amdgcn_wave_barrier();
        for (int c0 = 0; c0 < total; c0 += 64) {           // one call site for the exact selection
            const int c = c0 + lane;
            sel.offer_key(c < total ? cand[wave][c] : kMaxKey, c < total);
        }
    } else {
        // heavy ties at the cut: stream the row through the running selection instead
        for (int j0 = 0; j0 < n4; j0 += 64) {
            const int j4 = j0 + lane;
            const bool valid = j4 < n4;
            const float4 cur = row4[min(j4, n4 - 1)];
            sel.template offer<false>(cur.x, (uint32_t)(4 * j4 + 0), valid);
            sel.template offer<false>(cur.y, (uint32_t)(4 * j4 + 1), valid);
            sel.template offer<false>(cur.z, (uint32_t)(4 * j4 + 2), valid);
            sel.template offer<false>(cur.w, (uint32_t)(4 * j4 + 3), valid);
        }
    }
    sel.flush();
    if (lane < nprobe) {
        const u64 key = sel.best[0];
        const bool miss = key == kMaxKey;
        cdis[q * nprobe + lane] = miss ? FLT_MAX_F : ordered_to_f32((uint32_t)(key >> 32));
        keys[q * nprobe + lane] = miss ? -1 : (int64_t)(uint32_t)key;
    }
}

// Two-level select for wide rows (nlist > 8192): with the tile minima of the distance kernel, the
// nprobe-th smallest tile minimum T bounds the nprobe-th smallest distance from above (nprobe tiles
// hold an element <= T each), and a tile whose minimum is above T holds no element <= T.  So only
// the tiles with minimum <= T -- nprobe of them unless minima tie -- are read: 256 B per tile
// instead of the whole row, and the exact (distance, column) selection runs over those.
template <int KPL>
__global__ __launch_bounds__(256) void coarse_select_tiled_kernel(const float* __restrict__ dist,
                                                                  const float* __restrict__ tmin, int64_t nq,
                                                                  int nlist, int nprobe, float* __restrict__ cdis,
                                                                  int64_t* __restrict__ keys) {
    __shared__ u64 queue[4][64];
    __shared__ uint16_t qual[4][2048];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;   // whole wave; no workgroup barrier below
    const int ntiles = nlist >> 6;
    const float* tm = tmin + q * ntiles;
    WaveSelect<KPL> sel;
    sel.init(nprobe, queue[wave], lane);
    for (int j0 = 0; j0 < ntiles; j0 += 64) {
        const int j = j0 + lane;
        sel.offer(tm[min(j, ntiles - 1)], (uint32_t)j, j < ntiles);
    }
    sel.flush();
    const float T = sel.thr_own;           // FLT_MAX when fewer than nprobe tiles hold anything admissible
    int nqual = 0;
    for (int j0 = 0; j0 < ntiles; j0 += 64) {
        const int j = j0 + lane;
        const float v = tm[min(j, ntiles - 1)];
        const bool ok = j < ntiles && v <= T && v < FLT_MAX_F;
        const u64 mask = __ballot(ok);
        if (ok) qual[wave][nqual + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)j;
        nqual += __popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    sel.init(nprobe, queue[wave], lane);
    const float* row = dist + q * nlist;
    constexpr int PF = 8;                  // tiles in flight per wave
    for (int b = 0; b < nqual; b += PF) {
        float v[PF];
        int tile[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            tile[u] = qual[wave][min(b + u, nqual - 1)];
            v[u] = row[tile[u] * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (b + u < nqual) sel.offer(v[u], (uint32_t)(tile[u] * 64 + lane), true);   // ascending positions
    }
    sel.flush();
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = r * 64 + lane;
        if (e < nprobe) {
            const u64 key = sel.best[r];
            const bool miss = key == kMaxKey;
            cdis[q * nprobe + e] = miss ? FLT_MAX_F : ordered_to_f32((uint32_t)(key >> 32));
            keys[q * nprobe + e] = miss ? -1 : (int64_t)(uint32_t)key;
        }
    }
}

void launch_coarse_select(const float* dist, int64_t nq, int nlist, int nprobe, float* cdis,
                          int64_t* keys, hipStream_t s, const float* tmin) {
    if (nq <= 0) return;
    dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    if (tmin) {
        if (nprobe <= 64)
            hipLaunchKernelGGL(coarse_select_tiled_kernel<1>, grid, block, 0, s, dist, tmin, nq, nlist, nprobe, cdis, keys);
        else if (nprobe <= 256)
            hipLaunchKernelGGL(coarse_select_tiled_kernel<4>, grid, block, 0, s, dist, tmin, nq, nlist, nprobe, cdis, keys);
        else
            hipLaunchKernelGGL(coarse_select_tiled_kernel<16>, grid, block, 0, s, dist, tmin, nq, nlist, nprobe, cdis, keys);
        return;
    }
    if (nprobe <= 64 && (nlist & 3) == 0 && nlist >= 256 && nlist <= 8192) {
        if (nlist <= 1024)
            hipLaunchKernelGGL(coarse_select_reg_kernel<4>, grid, block, 0, s, dist, nq, nlist, nprobe, cdis, keys);
        else if (nlist <= 2048)
            hipLaunchKernelGGL(coarse_select_reg_kernel<8>, grid, block, 0, s, dist, nq, nlist, nprobe, cdis, keys);
        else if (nlist <= 4096)
            hipLaunchKernelGGL(coarse_select_reg_kernel<16>, grid, block, 0, s, dist, nq, nlist, nprobe, cdis, keys);
        else
            hipLaunchKernelGGL(coarse_select_reg_kernel<32>, grid, block, 0, s, dist, nq, nlist, nprobe, cdis, keys);
    } else if (nprobe <= 64)
        hipLaunchKernelGGL(coarse_select_kernel<1>, grid, block, 0, s, dist, nq, nlist, nprobe, cdis, keys);
    else if (nprobe <= 256)
        hipLaunchKernelGGL(coarse_select_kernel<4>, grid, block, 0, s, dist, nq, nlist, nprobe, cdis, keys);
    else
        hipLaunchKernelGGL(coarse_select_kernel<16>, grid, block, 0, s, dist, nq, nlist, nprobe, cdis, keys);
}

// ---------------------------------------------------------------------------
// PQ tables: out[v][m][j] for a tile of VT vectors x one sub-quantizer per
// workgroup.  The sub-quantizer's ksub centroids sit in LDS with a padded row
// (dsub+1) so that consecutive j hit distinct banks.
// ---------------------------------------------------------------------------
constexpr int kTableVT = 32;

// blockIdx.z: chunk of kch centroids (a 16 384-entry multi-index half table does not fit LDS at once)
__global__ __launch_bounds__(256) void pq_tables_kernel(
    const float* __restrict__ x, int64_t nv, int d, const float* __restrict__ cent, int M,
    int ksub, int dsub, const float* __restrict__ rnorm, int mode, float* __restrict__ out, int kch) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ds1 = dsub + 1;
    float* scent = sm;                  // [kch][dsub+1]
    float* sx = sm + kch * ds1;         // [VT][dsub]
    const int m = blockIdx.y;
    const int j0 = blockIdx.z * kch;
    const int nj = min(kch, ksub - j0);
    const int64_t v0 = (int64_t)blockIdx.x * kTableVT;
    const int t = threadIdx.x;
    const float* cm = cent + ((size_t)m * ksub + j0) * dsub;
    for (int e = t; e < nj * dsub; e += 256) scent[(e / dsub) * ds1 + (e % dsub)] = cm[e];
    for (int e = t; e < kTableVT * dsub; e += 256) {
        const int64_t v = v0 + e / dsub;
        sx[e] = v < nv ? x[v * d + m * dsub + (e % dsub)] : 0.f;
    }
    __syncthreads();
    for (int e = t; e < kTableVT * nj; e += 256) {
        const int vl = e / nj, j = e % nj;
        const int64_t v = v0 + vl;
        if (v >= nv) break;
        const float* xv = sx + vl * dsub;
        const float* cj = scent + j * ds1;
        float r;
        if (mode == 1) {
            r = l2sqr_sse_order([&](int c) { return xv[c]; }, [&](int c) { return cj[c]; }, dsub);
        } else {
            r = ip_sse_order([&](int c) { return xv[c]; }, [&](int c) { return cj[c]; }, dsub);
            if (mode == 2)  // fvec_madd(r_norms, 2.0, tab): a + bf*b (utils.cpp:1832-1853)
                r = __fadd_rn(rnorm[m * ksub + j0 + j], __fmul_rn(2.f, r));
        }
        out[((size_t)v * M + m) * ksub + j0 + j] = r;
    }
}

void launch_pq_tables(const float* x, int64_t nv, int d, const float* cent, int M, int ksub,
                      int dsub, const float* rnorm, int mode, float* out, hipStream_t s) {
    if (nv <= 0) return;
    // centroids of one chunk + the vectors' sub-vectors in at most 64 KB of LDS
    int kch = ksub;
    while ((size_t)kch * (dsub + 1) * sizeof(float) > 60 * 1024 && kch > 64) kch = (kch + 1) / 2;
    const size_t smem = ((size_t)kch * (dsub + 1) + (size_t)kTableVT * dsub) * sizeof(float);
    ensure_dynamic_lds(reinterpret_cast<const void*>(pq_tables_kernel), smem);
    dim3 grid((unsigned)((nv + kTableVT - 1) / kTableVT), (unsigned)M, (unsigned)((ksub + kch - 1) / kch));
    hipLaunchKernelGGL(pq_tables_kernel, grid, dim3(256), smem, s, x, nv, d, cent, M, ksub, dsub,
                       rnorm, mode, out, kch);
}

// ---------------------------------------------------------------------------
// list scan.  One 256-thread workgroup (4 waves) per query:
//   * the per-query table part stays in registers for the whole query
//     (FAST path: 16 entries per thread, already multiplied by -2);
//   * per probe: term2[key] (16 KB at M=16,ksub=256) is prefetched from L2/MALL one
//     probe ahead, combined as  t2 + (-2*ip)  -- the exact fvec_madd value,
//     IndexIVFPQ.cpp:641-644 -- and written to one of two LDS LUT buffers, so
//     building LUT p+1 overlaps scanning with LUT p and one barrier per probe
//     suffices;
//   * the scan streams the list's packed codes with one 16-byte load per lane
//     (coalesced 1 KiB per wave-instruction) and sums  dis0 + t[0] + ... + t[M-1]
//     strictly left to right (IndexIVFPQ.cpp:788-794);
//   * each wave keeps a running selection (wave_topk.cuh); the four are merged
//     at the end and the winners' ids are fetched (ids are never touched in the
//     scan itself).
// ---------------------------------------------------------------------------
template <int KPL, bool FAST16>
__global__ __launch_bounds__(256) void scan_kernel(ScanArgs a, int nbuf, int lut_region) {
    // LDS: [lut_region bytes: LUT buffers, later the 4 x k merge area][queue][cum][residual]
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int E = a.M * a.ksub;
    float* lut = reinterpret_cast<float*>(smraw);                       // [nbuf][E]
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);            // [4][64]
    uint32_t* cum = reinterpret_cast<uint32_t*>(queue + 4 * 64);        // [nprobe+1]
    float* sres = reinterpret_cast<float*>(cum + a.nprobe + 1);         // [d] (table mode 0)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t q = blockIdx.x;
    const int64_t* kq = a.keys + q * a.nprobe;
    const float* cq = a.coarse_dis + q * a.nprobe;
    const float* qt = a.qtab ? a.qtab + q * E : nullptr;

    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);

    // FAST16: 4 float4 per thread of -2 * ip_table (mode 1)
    float4 m2t3[4];
    float4 t2r[4];
    if (FAST16) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float4 v = reinterpret_cast<const float4*>(qt)[i * 256 + t];
            m2t3[i] = make_float4(__fmul_rn(-2.f, v.x), __fmul_rn(-2.f, v.y), __fmul_rn(-2.f, v.z),
                                  __fmul_rn(-2.f, v.w));
        }
        const int64_t key0 = a.nprobe > 0 ? kq[0] : -1;
        if (key0 >= 0 && key0 < a.nlist) {
            const float4* src = reinterpret_cast<const float4*>(a.term2 + key0 * E);
#pragma unroll
            for (int i = 0; i < 4; i++) t2r[i] = src[i * 256 + t];
        }
    } else if (a.table_mode == 2) {
        // not by_residual: one distance table per query (IndexIVFPQ.cpp:558-559)
        for (int e = t; e < E; e += 256) lut[e] = qt[e];
        __syncthreads();
    }

    uint32_t pos0 = 0;
    int64_t nscan = 0;
    int buf = 0;
    int ik = 0;
    bool badkey = false;
    for (; ik < a.nprobe; ik++) {
        const int64_t key = kq[ik];
        if (t == 0) cum[ik] = pos0;
        int64_t len = 0, off = 0;
        if (key >= a.nlist) badkey = true;                  // IndexIVFPQ.cpp:1008-1011
        const bool live = key >= 0 && key < a.nlist;
        if (live) { off = a.list_off[key]; len = a.list_len ? a.list_len[key] : a.list_off[key + 1] - off; }
        float dis0 = 0.f;

        if (len > 0) {
            float* L = lut + (size_t)buf * E;
            if (FAST16) {
                dis0 = cq[ik];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float4 s;
                    s.x = __fadd_rn(t2r[i].x, m2t3[i].x);
                    s.y = __fadd_rn(t2r[i].y, m2t3[i].y);
                    s.z = __fadd_rn(t2r[i].z, m2t3[i].z);
                    s.w = __fadd_rn(t2r[i].w, m2t3[i].w);
                    reinterpret_cast<float4*>(L)[i * 256 + t] = s;
                }
            } else if (a.table_mode == 1) {
                dis0 = cq[ik];
                if (a.imi_nbits > 0) {
                    // table type 2 (IndexIVFPQ.cpp:645-686): sub-quantizer m takes its row from the
                    // coarse sub-index of the half it belongs to
                    const int Mf = a.M / 2;
                    const int64_t ki0 = key & ((int64_t(1) << a.imi_nbits) - 1), ki1 = key >> a.imi_nbits;
                    for (int e = t; e < E; e += 256) {
                        const int64_t ki = (e / a.ksub) < Mf ? ki0 : ki1;
                        L[e] = __fadd_rn(a.term2[ki * E + e], __fmul_rn(-2.f, qt[e]));
                    }
                } else {
                    const float* t2 = a.term2 + key * E;
                    for (int e = t; e < E; e += 256)
                        L[e] = __fadd_rn(t2[e], __fmul_rn(-2.f, qt[e]));
                }
            } else if (a.table_mode == 0) {
                // residual tables: compute_residual + compute_distance_table
                // (IndexIVFPQ.cpp:636-637)
                const float* c = a.coarse + key * a.d;
                const float* qv = a.queries + q * a.d;
                for (int e = t; e < a.d; e += 256) sres[e] = __fsub_rn(qv[e], c[e]);
                __syncthreads();
                for (int e = t; e < E; e += 256) {
                    const int m = e / a.ksub;
                    const float* xv = sres + m * a.dsub;
                    const float* cj = a.pq_cent + (size_t)e * a.dsub;
                    L[e] = l2sqr_sse_order([&](int c2) { return xv[c2]; },
                                           [&](int c2) { return cj[c2]; }, a.dsub);
                }
            }
        }
        if (FAST16 && ik + 1 < a.nprobe) {
            const int64_t kn = kq[ik + 1];
            if (kn >= 0 && kn < a.nlist) {
                const float4* src = reinterpret_cast<const float4*>(a.term2 + kn * E);
#pragma unroll
                for (int i = 0; i < 4; i++) t2r[i] = src[i * 256 + t];
            }
        }
        if (len > 0) {
            if (a.table_mode != 2) __syncthreads();
            const float* L = a.table_mode == 2 ? lut : lut + (size_t)buf * E;
            if (FAST16) {
                const uint4* cp = reinterpret_cast<const uint4*>(a.codes + off * 16);
                for (int64_t j0 = (int64_t)wave * 64; j0 < len; j0 += 256) {
                    const int64_t j = j0 + lane;
                    const bool valid = j < len;
                    float dis = 0.f;
                    if (valid) {
                        const uint4 c = cp[j];
                        dis = dis0;
                        dis = __fadd_rn(dis, L[0 * 256 + (c.x & 255u)]);
                        dis = __fadd_rn(dis, L[1 * 256 + ((c.x >> 8) & 255u)]);
                        dis = __fadd_rn(dis, L[2 * 256 + ((c.x >> 16) & 255u)]);
                        dis = __fadd_rn(dis, L[3 * 256 + (c.x >> 24)]);
                        dis = __fadd_rn(dis, L[4 * 256 + (c.y & 255u)]);
                        dis = __fadd_rn(dis, L[5 * 256 + ((c.y >> 8) & 255u)]);
                        dis = __fadd_rn(dis, L[6 * 256 + ((c.y >> 16) & 255u)]);
                        dis = __fadd_rn(dis, L[7 * 256 + (c.y >> 24)]);
                        dis = __fadd_rn(dis, L[8 * 256 + (c.z & 255u)]);
                        dis = __fadd_rn(dis, L[9 * 256 + ((c.z >> 8) & 255u)]);
                        dis = __fadd_rn(dis, L[10 * 256 + ((c.z >> 16) & 255u)]);
                        dis = __fadd_rn(dis, L[11 * 256 + (c.z >> 24)]);
                        dis = __fadd_rn(dis, L[12 * 256 + (c.w & 255u)]);
                        dis = __fadd_rn(dis, L[13 * 256 + ((c.w >> 8) & 255u)]);
                        dis = __fadd_rn(dis, L[14 * 256 + ((c.w >> 16) & 255u)]);
                        dis = __fadd_rn(dis, L[15 * 256 + (c.w >> 24)]);
                    }
                    sel.offer(dis, pos0 + (uint32_t)j, valid);
                }
            } else {
                const uint8_t* cp = a.codes + off * a.M;
                for (int64_t j0 = (int64_t)wave * 64; j0 < len; j0 += 256) {
                    const int64_t j = j0 + lane;
                    const bool valid = j < len;
                    float dis = 0.f;
                    if (valid) {
                        const uint8_t* cj = cp + j * a.M;
                        dis = dis0;
                        const float* tab = L;
                        if ((a.M & 3) == 0) {
                            const uint32_t* cw = reinterpret_cast<const uint32_t*>(cj);
                            for (int w = 0; w < a.M / 4; w++) {
                                const uint32_t c = cw[w];
                                dis = __fadd_rn(dis, tab[c & 255u]); tab += a.ksub;
                                dis = __fadd_rn(dis, tab[(c >> 8) & 255u]); tab += a.ksub;
                                dis = __fadd_rn(dis, tab[(c >> 16) & 255u]); tab += a.ksub;
                                dis = __fadd_rn(dis, tab[c >> 24]); tab += a.ksub;
                            }
                        } else {
                            for (int m = 0; m < a.M; m++) {
                                dis = __fadd_rn(dis, tab[cj[m]]);
                                tab += a.ksub;
                            }
                        }
                    }
                    sel.offer(dis, pos0 + (uint32_t)j, valid);
                }
            }
            if (nbuf == 2) buf ^= 1;
            else if (a.table_mode != 2) __syncthreads();   // single buffer: scan done before rebuild
        }
        nscan += len;
        pos0 += (uint32_t)len;
        if (a.max_codes && nscan >= a.max_codes) { ik++; break; }   // IndexIVFPQ.cpp:1033
    }
    if (t == 0)
        for (int i = ik; i <= a.nprobe; i++) cum[i] = pos0;

    // ---- merge the four waves' selections, emit ----
    merge_and_emit<KPL>(sel, smraw, cum, a, q, wave, lane,
                        [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = a.list_off[lkey]; });
    if (t == 0) {
        atomicAdd(a.ncode, (unsigned long long)nscan);
        if (badkey) *a.bad_key = 1;
    }
}

template <int KPL, bool FAST16>
static void launch_scan_t(const ScanArgs& a, int nbuf, int lut_region, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan_kernel<KPL, FAST16>), smem);
    hipLaunchKernelGGL((scan_kernel<KPL, FAST16>), dim3((unsigned)a.nq), dim3(256), smem, s, a, nbuf,
                       lut_region);
}

void launch_scan(const ScanArgs& a, hipStream_t s) {
    if (a.nq <= 0) return;
    const size_t E = (size_t)a.M * a.ksub;
    const size_t tail = 4 * 64 * 8 + ((size_t)a.nprobe + 1) * 4 + (size_t)a.d * 4 + 16;
    int nbuf = (a.table_mode == 2) ? 1 : 2;
    if (2 * E * 4 + tail > 150 * 1024) nbuf = 1;
    size_t lutb = (size_t)nbuf * E * 4;
    const size_t merge = (size_t)4 * a.k * 8;       // merge area aliases the LUT buffers
    if (lutb < merge) lutb = merge;
    lutb = (lutb + 15) & ~(size_t)15;
    const int lut_region = (int)lutb;
    const size_t smem = lutb + tail;
    if (a.k <= 64) launch_scan_t<1, false>(a, nbuf, lut_region, smem, s);
    else if (a.k <= 256) launch_scan_t<4, false>(a, nbuf, lut_region, smem, s);
    else launch_scan_t<16, false>(a, nbuf, lut_region, smem, s);
}

// ---------------------------------------------------------------------------
// merge of per-shard top-k lists (GpuIndexIVFPQ::merge / mergekernel,
// gpu/GpuIndexIVFPQ.cu:1467-1591; IndexShards merge_tables, MetaIndexes.cpp:486-557):
// parts are [nparts][nq][k]; one wave per query selects the k smallest
// (distance, part*k + rank).
// ---------------------------------------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ Dp,
                                                         const int64_t* __restrict__ Ip, int64_t nq,
                                                         int k, int nparts, float* __restrict__ D,
                                                         int64_t* __restrict__ I, const int* __restrict__ row_map) {
    __shared__ u64 queue[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;
    // row_map (the split tail of a batch, scan16.hip): partial row q belongs to output row row_map[q]; -1: unused slot
    const int64_t orow = row_map ? (int64_t)row_map[q] : q;
    if (orow < 0) return;
    WaveSelect<KPL> sel;
    sel.init(k, queue[wave], lane);
    const int num = nparts * k;
    for (int i0 = 0; i0 < num; i0 += 64) {
        const int i = i0 + lane;
        bool valid = i < num;
        float v = 0.f;
        if (valid) {
            const int64_t src = ((int64_t)(i / k) * nq + q) * k + (i % k);
            // padding entries are (FLT_MAX, -1): the strict admission below never takes FLT_MAX, and a
            // real result is always < FLT_MAX (Heap.h:76-78), so padding is recognised by its distance --
            // NOT by id < 0: callers may store negative ids (add_with_ids)
            v = Dp[src];
        }
        sel.offer(v, (uint32_t)i, valid);
    }
    sel.flush();
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = r * 64 + lane;
        if (e >= k) continue;
        const u64 key = sel.best[r];
        float dis = FLT_MAX_F;
        int64_t id = -1;
        if (key != kMaxKey) {
            const int i = (int)(uint32_t)key;
            const int64_t src = ((int64_t)(i / k) * nq + q) * k + (i % k);
            dis = Dp[src];
            id = Ip[src];
        }
        D[orow * k + e] = dis;
        I[orow * k + e] = id;
    }
}

void launch_merge_topk(const float* Dp, const int64_t* Ip, int64_t nq, int k, int nparts, float* D,
                       int64_t* I, hipStream_t s, const int* row_map) {
    if (nq <= 0) return;
    dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    if (k <= 64) hipLaunchKernelGGL(merge_topk_kernel<1>, grid, block, 0, s, Dp, Ip, nq, k, nparts, D, I, row_map);
    else if (k <= 256) hipLaunchKernelGGL(merge_topk_kernel<4>, grid, block, 0, s, Dp, Ip, nq, k, nparts, D, I, row_map);
    else if (k <= 512) hipLaunchKernelGGL(merge_topk_kernel<8>, grid, block, 0, s, Dp, Ip, nq, k, nparts, D, I, row_map);
    else hipLaunchKernelGGL(merge_topk_kernel<16>, grid, block, 0, s, Dp, Ip, nq, k, nparts, D, I, row_map);
}

__global__ void gather_cols_kernel(const float* __restrict__ x, int64_t n, int d, int col0, int dc,
                                   float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * dc) return;
    out[e] = x[(e / dc) * d + col0 + (e % dc)];
}

void launch_gather_cols(const float* x, int64_t n, int d, int col0, int dc, float* out, hipStream_t s) {
    if (n <= 0) return;
    const int64_t tot = n * dc;
    hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, x, n, d,
                       col0, dc, out);
}

// ---------------------------------------------------------------------------
// MinSumK replay (IndexPQ.cpp:690-778), M = 2 terms: the k smallest sums v0[r0] + v1[r1]
// in the order and with the float values the reference's heap walk produces: the first sum
// is (0 + v0[0]) + v1[0]; every later one is its predecessor's sum plus the DIFFERENCE of
// consecutive sorted table entries, so the values are path dependent and have to be
// replayed step by step.  One thread per query; binary min-heap (Heap.h:89-143 with CMin)
// of at most 2k entries in a global scratch row.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void minheap_push_dev(int k, float* bh_val, int64_t* bh_ids, float val, int64_t id) {
    bh_val--; bh_ids--;
    int i = k, i_father;
    while (i > 1) {
        i_father = i >> 1;
        if (!(val < bh_val[i_father])) break;
        bh_val[i] = bh_val[i_father]; bh_ids[i] = bh_ids[i_father]; i = i_father;
    }
    bh_val[i] = val; bh_ids[i] = id;
}
__device__ __forceinline__ void minheap_pop_dev(int k, float* bh_val, int64_t* bh_ids) {
    bh_val--; bh_ids--;
    const float val = bh_val[k];
    int i = 1, i1, i2;
    while (1) {
        i1 = i << 1; i2 = i1 + 1;
        if (i1 > k) break;
        if (i2 == k + 1 || bh_val[i1] < bh_val[i2]) {
            if (val < bh_val[i1]) break;
            bh_val[i] = bh_val[i1]; bh_ids[i] = bh_ids[i1]; i = i1;
        } else {
            if (val < bh_val[i2]) break;
            bh_val[i] = bh_val[i2]; bh_ids[i] = bh_ids[i2]; i = i2;
        }
    }
    bh_val[i] = bh_val[k]; bh_ids[i] = bh_ids[k];
}

__global__ void imi_minsum_kernel(const float* __restrict__ sv0, const int64_t* __restrict__ si0,
                                  const float* __restrict__ sv1, const int64_t* __restrict__ si1, int T,
                                  int64_t nq, int k, int kc, int imi_nbits, float* __restrict__ heap_val,
                                  int64_t* __restrict__ heap_id, float* __restrict__ sums,
                                  int64_t* __restrict__ keys) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float* v0 = sv0 + q * T;
    const float* v1 = sv1 + q * T;
    const int64_t* i0 = si0 + q * T;
    const int64_t* i1 = si1 + q * T;
    float* out_s = sums + q * k;
    int64_t* out_k = keys + q * k;
    if (k == 1) {   // IndexPQ.cpp:815-840: minimum of each table (first minimum), dis = (0 + m0) + m1
        out_s[0] = __fadd_rn(__fadd_rn(0.f, v0[0]), v1[0]);
        out_k[0] = i0[0] | (i1[0] << imi_nbits);
        return;
    }
    float* hv = heap_val + q * 2 * k;
    int64_t* hi = heap_id + q * 2 * k;
    int hs = 0;
    // terms are encoded as r0 + r1 * kc over the RANKS; translated to indices on output
    float sum = __fadd_rn(__fadd_rn(0.f, v0[0]), v1[0]);
    out_s[0] = sum;
    out_k[0] = i0[0] | (i1[0] << imi_nbits);
    if (T > 1) {
        minheap_push_dev(++hs, hv, hi, __fadd_rn(sum, __fsub_rn(v0[1], v0[0])), 1);
        minheap_push_dev(++hs, hv, hi, __fadd_rn(sum, __fsub_rn(v1[1], v1[0])), (int64_t)kc);
    }
    for (int kk = 1; kk < k; kk++) {
        if (hs == 0) { out_s[kk] = 3.402823466e+38f; out_k[kk] = -1; continue; }   // fewer than k cells
        const float s2 = hv[0];
        const int64_t ti = hi[0];
        const int r0 = (int)(ti % kc), r1 = (int)(ti / kc);
        out_s[kk] = s2;
        out_k[kk] = i0[r0] | (i1[r1] << imi_nbits);
        do { minheap_pop_dev(hs--, hv, hi); } while (hs > 0 && hi[0] == ti);
        if (r0 + 1 < kc && r0 + 1 < T)
            minheap_push_dev(++hs, hv, hi, __fadd_rn(s2, __fsub_rn(v0[r0 + 1], v0[r0])), ti + 1);
        if (r1 + 1 < kc && r1 + 1 < T)
            minheap_push_dev(++hs, hv, hi, __fadd_rn(s2, __fsub_rn(v1[r1 + 1], v1[r1])), ti + kc);
    }
}

// The same walk with the heap and the two sorted tables in LDS (element j of thread t at
// [j * NTH + t]): the global-memory heap above pays a memory round trip per heap level, 64 pops
// x ~7 levels x 2 arrays, and was 475 us of a 660 us multi-index coarse stage at nprobe = 64.
// Term indices fit 32 bits (kc^2 <= 2^30).  Same comparisons, same float operations.
template <int NTH>
__global__ __launch_bounds__(NTH) void imi_minsum_lds_kernel(
    const float* __restrict__ sv0, const int64_t* __restrict__ si0, const float* __restrict__ sv1,
    const int64_t* __restrict__ si1, int T, int64_t nq, int k, int kc, int imi_nbits,
    float* __restrict__ sums, int64_t* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int t = threadIdx.x;
    // heap entry = one 64-bit LDS word {float bits : term index}: one access moves both
    unsigned long long* hq = reinterpret_cast<unsigned long long*>(smraw) + t;   // [2k][NTH]
    float* v0 = reinterpret_cast<float*>(smraw) + 4 * k * NTH + t;               // [T][NTH]
    float* v1 = v0 + T * NTH;
    // the sub-quantizer indices behind the ranks, in LDS as well: no global load inside the walk (218 -> 210 us at
    // 10 000 queries, nprobe 64: the walk is bound by its ~35 dependent LDS round trips per emitted cell, not by this)
    int32_t* x0 = reinterpret_cast<int32_t*>(v1 + T * NTH);                    // [T][NTH]
    int32_t* x1 = x0 + T * NTH;
    const int64_t q = (int64_t)blockIdx.x * NTH + t;
    if (q >= nq) return;
    for (int j = 0; j < T; j++) {
        v0[j * NTH] = sv0[q * T + j]; v1[j * NTH] = sv1[q * T + j];
        x0[j * NTH] = (int32_t)si0[q * T + j]; x1[j * NTH] = (int32_t)si1[q * T + j];
    }
    float* out_s = sums + q * k;
    int64_t* out_k = keys + q * k;
    auto fval = [](unsigned long long e) { return __uint_as_float((uint32_t)(e >> 32)); };
    auto push = [&](int n, float val, int32_t id) {     // Heap.h:110-127 (min-heap), 1-based
        int i = n;
        while (i > 1) {
            const int f = i >> 1;
            const unsigned long long ef = hq[(f - 1) * NTH];
            if (!(val < fval(ef))) break;
            hq[(i - 1) * NTH] = ef; i = f;
        }
        hq[(i - 1) * NTH] = ((unsigned long long)__float_as_uint(val) << 32) | (uint32_t)id;
    };
    auto pop = [&](int n) {                             // Heap.h:89-108
        const unsigned long long last = hq[(n - 1) * NTH];
        const float val = fval(last);
        int i = 1;
        while (1) {
            const int i1c = i << 1, i2c = i1c + 1;
            if (i1c > n) break;
            const unsigned long long e1 = hq[(i1c - 1) * NTH];
            const unsigned long long e2 = i2c <= n ? hq[(i2c - 1) * NTH] : 0ull;
            if (i2c == n + 1 || fval(e1) < fval(e2)) {
                if (val < fval(e1)) break;
                hq[(i - 1) * NTH] = e1; i = i1c;
            } else {
                if (val < fval(e2)) break;
                hq[(i - 1) * NTH] = e2; i = i2c;
            }
        }
        hq[(i - 1) * NTH] = last;
    };
    // a term (r0, r1) travels as r0 | r1 << 16 instead of the reference's r0 + r1 * kc: the walk only ever tests two
    // terms for equality, and the ranks come back without a division (kc <= 32768, checked by the launcher)
    int hs = 0;
    const float sum = __fadd_rn(__fadd_rn(0.f, v0[0]), v1[0]);
    out_s[0] = sum;
    out_k[0] = (int64_t)x0[0] | ((int64_t)x1[0] << imi_nbits);
    if (T > 1) {
        push(++hs, __fadd_rn(sum, __fsub_rn(v0[1 * NTH], v0[0])), 1);
        push(++hs, __fadd_rn(sum, __fsub_rn(v1[1 * NTH], v1[0])), 1 << 16);
    }
    for (int kk = 1; kk < k; kk++) {
        if (hs == 0) { out_s[kk] = 3.402823466e+38f; out_k[kk] = -1; continue; }
        const unsigned long long top = hq[0];
        const float s2 = fval(top);
        const int32_t ti = (int32_t)(uint32_t)top;
        const int r0 = ti & 0xffff, r1 = ti >> 16;
        out_s[kk] = s2;
        out_k[kk] = (int64_t)x0[r0 * NTH] | ((int64_t)x1[r1 * NTH] << imi_nbits);
        do { pop(hs--); } while (hs > 0 && (int32_t)(uint32_t)hq[0] == ti);
        if (r0 + 1 < kc && r0 + 1 < T)
            push(++hs, __fadd_rn(s2, __fsub_rn(v0[(r0 + 1) * NTH], v0[r0 * NTH])), ti + 1);
        if (r1 + 1 < kc && r1 + 1 < T)
            push(++hs, __fadd_rn(s2, __fsub_rn(v1[(r1 + 1) * NTH], v1[r1 * NTH])), ti + (1 << 16));
    }
}

// The same walk by ONE WAVE per query with the heap in registers (round 5; k <= 64 and T <= 64: the heap never holds more than
// 64 entries when a pop or a push needs it, see below).  The thread-per-query kernels above spend ~35 DEPENDENT memory round
// trips per emitted cell and keep 313 half-empty waves on the chip for 10 000 queries: 209 us at nprobe 64, 41 % of the
// multi-index coarse stage.  Here lane i holds heap node i + 1 and table entries v0[i], v1[i], x0[i], x1[i]; lane kk collects
// output kk.  Heap.h:89-127's sift loops become
//   push: the new slot's ancestors all compare with the value at once; the deepest ancestor that does not lose stops the
//         sift, every path node below it takes its parent's entry, the topmost of them the new one;
//   pop:  every node picks its smaller child at once (the reference's rule, incl. `i2 == k + 1`); a scalar walk from the root
//         follows the picks while the last entry does not win; the nodes walked take their picked child's entry, the end of the
//         walk the last one.
// Same comparisons on the same values in the same heap positions => the same pops in the same order, ties included.
// Measured (10 000 queries, 2 x 14 bits, nprobe 64): 209 -> 168 us with a scalar walk in the pop (~10 instructions per heap
// level), 129 us since round 6 (the path by ballots, below).  Bound by instruction issue (~120 wave instructions per emitted cell).
// Size: two entries after the first cell, at most one more per emitted cell => at most kk + 1 before the pops of iteration kk
// and kk + 2 after its pushes; the pushes of the last iteration feed nothing and are skipped, so <= 64 for k <= 64.
template <int NW>
__global__ __launch_bounds__(64 * NW) void imi_minsum_wave_kernel(
    const float* __restrict__ sv0, const int64_t* __restrict__ si0, const float* __restrict__ sv1,
    const int64_t* __restrict__ si1, int T, int64_t nq, int k, int kc, int imi_nbits,
    float* __restrict__ sums, int64_t* __restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
    if (q >= nq) return;
    const bool in_t = lane < T;
    const float v0 = in_t ? sv0[q * T + lane] : 0.f, v1 = in_t ? sv1[q * T + lane] : 0.f;
    const int x0 = in_t ? (int)si0[q * T + lane] : 0, x1 = in_t ? (int)si1[q * T + lane] : 0;
    auto rlf = [](float x, int l) __attribute__((always_inline)) {      // l wave-uniform
        return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), l));
    };
    auto rli = [](int x, int l) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(x, l); };
    float hv = 0.f;        // heap node lane + 1: value
    int hid = 0;           // ... term (r0 | r1 << 16, as in the LDS kernel)
    int n = 0;             // heap size (wave-uniform)
    float os = 3.402823466e+38f;     // output slot `lane`: the sum and the term it belongs to (translated to a key at the end)
    int oterm = -1;
    const int pos = lane + 1;
    // byte addresses of the lane permutations (ds_bpermute): parent, both children
    const int a_par = ((pos >> 1) - 1) * 4, a_c1 = (2 * pos - 1) * 4, a_c2 = (2 * pos) * 4;
    auto perm_f = [](int addr, float x) __attribute__((always_inline)) {
        return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(x)));
    };
    auto push = [&](float val, int id) __attribute__((always_inline)) {
        n++;
        const int sh = __clz(pos) - __clz(n);                  // levels between this node and the new slot
        const bool onp = sh >= 0 && (n >> sh) == pos;          // on the new slot's root path (the slot itself: sh = 0)
        const u64 stopm = __ballot(onp && sh >= 1 && !(val < hv));
        const int stop_pos = stopm ? 64 - (int)__clzll(stopm) : 0;      // deepest ancestor that stays (heap index; 0: none)
        const float pv = perm_f(a_par, hv);
        const int pid = __builtin_amdgcn_ds_bpermute(a_par, hid);
        if (onp && pos > stop_pos) {
            const bool topmost = (pos >> 1) == stop_pos;
            hv = topmost ? val : pv;
            hid = topmost ? id : pid;
        }
    };
    // (round 6) the walk from the root without a loop: a node is on the sift's path when it and all its ancestors below the root
    // are their parents' picks -- one ballot of "picked by my parent" against a per-lane CONSTANT mask of those ancestors --, the
    // walk ends at the first node of the path (lanes are in level order) that has no child or whose picked child beats the last
    // entry.  168 -> 129 us per 10 000 queries at 64 cells (the scalar walk: ~10 instructions per heap level with a readlane
    // and a branch each; round 5 built the ancestors' bits with six 64-bit shifts per pop: 177 us).
    u64 anc = 0;                                        // bits of this node and of its ancestors, the root excluded
    for (int a = pos; a >= 2; a >>= 1) anc |= 1ull << (a - 1);
    const int a_parent = ((pos >> 1) - 1) * 4;
    auto pop = [&]() __attribute__((always_inline)) {
        const float lastv = rlf(hv, n - 1);
        const int lastid = rli(hid, n - 1);
        const float cv1 = perm_f(a_c1, hv), cv2 = perm_f(a_c2, hv);
        const int ci1 = __builtin_amdgcn_ds_bpermute(a_c1, hid), ci2 = __builtin_amdgcn_ds_bpermute(a_c2, hid);
        const bool pick1 = (2 * pos + 1 == n + 1) || (cv1 < cv2);
        const float ccv = pick1 ? cv1 : cv2;
        const int ccid = pick1 ? ci1 : ci2;
        // am I my parent's pick?  (left children sit at even positions)
        const int ppick1 = __builtin_amdgcn_ds_bpermute(a_parent, pick1 ? 1 : 0);
        const bool mine = pos == 1 || ((pos & 1) == 0) == (ppick1 != 0);
        const u64 picked = __ballot(mine && pos <= n);
        const u64 on = __ballot((picked & anc) == anc && pos <= n);
        // the walk stops at a node without children, or whose picked child the last entry beats
        const u64 stop = __ballot(2 * pos > n || lastv < ccv) & on;
        const int cur = __builtin_ctzll(stop) + 1;          // (a path always ends: its deepest node has no child)
        const u64 walked = on & ((1ull << (cur - 1)) - 1ull);
        if ((walked >> lane) & 1ull) { hv = ccv; hid = ccid; }
        if (lane == cur - 1) { hv = lastv; hid = lastid; }
        n--;
    };
    const float sum = __fadd_rn(__fadd_rn(0.f, rlf(v0, 0)), rlf(v1, 0));
    if (lane == 0) { os = sum; oterm = 0; }
    if (T > 1 && k > 1) {
        push(__fadd_rn(sum, __fsub_rn(rlf(v0, 1), rlf(v0, 0))), 1);
        push(__fadd_rn(sum, __fsub_rn(rlf(v1, 1), rlf(v1, 0))), 1 << 16);
    }
    for (int kk = 1; kk < k; kk++) {
        if (n == 0) break;                                      // fewer than k cells: the remaining slots keep FLT_MAX / -1
        const float s2 = rlf(hv, 0);
        const int ti = rli(hid, 0);
        const int r0 = ti & 0xffff, r1 = ti >> 16;
        if (lane == kk) { os = s2; oterm = ti; }
        do { pop(); } while (n > 0 && rli(hid, 0) == ti);
        if (kk == k - 1) break;                                 // (these pushes would feed nothing)
        if (r0 + 1 < kc && r0 + 1 < T) push(__fadd_rn(s2, __fsub_rn(rlf(v0, r0 + 1), rlf(v0, r0))), ti + 1);
        if (r1 + 1 < kc && r1 + 1 < T) push(__fadd_rn(s2, __fsub_rn(rlf(v1, r1 + 1), rlf(v1, r1))), ti + (1 << 16));
    }
    // the sub-quantizer indices behind the ranks of every output at once
    const int tt = max(oterm, 0);
    const int k0 = __builtin_amdgcn_ds_bpermute((tt & 0xffff) * 4, x0), k1 = __builtin_amdgcn_ds_bpermute((tt >> 16) * 4, x1);
    const int64_t ok = oterm < 0 ? -1 : ((int64_t)k0 | ((int64_t)k1 << imi_nbits));
    if (lane < k) { sums[q * k + lane] = os; keys[q * k + lane] = ok; }
}

void launch_imi_minsum(const float* sv0, const int64_t* si0, const float* sv1, const int64_t* si1, int T,
                       int64_t nq, int k, int kc, int imi_nbits, float* heap_val, int64_t* heap_id,
                       float* sums, int64_t* keys, hipStream_t s) {
    if (nq <= 0) return;
    static const bool no_wave = getenv("VLQ_IMI_MINSUM_LDS") != nullptr;       // (A/B: the thread-per-query walk)
    if (k > 1 && k <= 64 && T <= 64 && kc <= 32768 && !no_wave) {
        constexpr int NW = 4;
        hipLaunchKernelGGL(imi_minsum_wave_kernel<NW>, dim3((unsigned)((nq + NW - 1) / NW)), dim3(64 * NW), 0, s,
                           sv0, si0, sv1, si1, T, nq, k, kc, imi_nbits, sums, keys);
        return;
    }
    // beyond 64 cells: one wave per query with its heap in LDS, the lanes sharing a sift's comparisons (imi_wide.hip; 10 000
    // queries, 2 x 14 bits, k = 128: 1.09 ms against the thread-per-query kernel's 1.55, whose 32 heaps per workgroup stop
    // fitting LDS at 128 and then live in global memory -- VLQ_IMI_MINSUM_WIDE_FROM=129 for the A/B)
    static const int wide_from = getenv("VLQ_IMI_MINSUM_WIDE_FROM") ? atoi(getenv("VLQ_IMI_MINSUM_WIDE_FROM")) : 65;
    if (k >= wide_from && imi_minsum_wide_ok(T, k, kc)) {
        launch_imi_minsum_wide(sv0, si0, sv1, si1, T, nq, k, kc, imi_nbits, sums, keys, s);
        return;
    }
    constexpr int NTH = 32;
    const size_t smem = (size_t)NTH * ((size_t)4 * k * 4 + (size_t)4 * T * 4);
    if (k > 1 && smem <= 128 * 1024 && kc <= 32768 && T <= 32768) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(imi_minsum_lds_kernel<NTH>), smem);
        hipLaunchKernelGGL(imi_minsum_lds_kernel<NTH>, dim3((unsigned)((nq + NTH - 1) / NTH)), dim3(NTH), smem, s,
                           sv0, si0, sv1, si1, T, nq, k, kc, imi_nbits, sums, keys);
        return;
    }
    hipLaunchKernelGGL(imi_minsum_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, s, sv0, si0, sv1,
                       si1, T, nq, k, kc, imi_nbits, heap_val, heap_id, sums, keys);
}

__global__ void transpose_pq_kernel(const float* __restrict__ in, int M, int ksub, int dsub,
                                    float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)M * ksub * dsub;
    if (e >= n) return;
    const int c = (int)(e % dsub);
    const int j = (int)((e / dsub) % ksub);
    const int m = (int)(e / ((int64_t)dsub * ksub));
    out[((int64_t)m * dsub + c) * ksub + j] = in[e];
}

void launch_transpose_pq(const float* in, int M, int ksub, int dsub, float* out, hipStream_t s) {
    const int64_t n = (int64_t)M * ksub * dsub;
    hipLaunchKernelGGL(transpose_pq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, M,
                       ksub, dsub, out);
}

// ---------------------------------------------------------------------------
// encode: residual to the assigned centroid, then per sub-quantizer the first
// minimum of fvec_L2sqr (ProductQuantizer.cpp:311-336: strict '<', mindis = 1e20).
// One wave per vector: lane j scans centroids j, j+64, ... and the wave reduces
// (distance, index) lexicographically, which is exactly "first minimum wins".
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void encode_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ coarse,
    const int64_t* __restrict__ assign, int by_residual, const float* __restrict__ cent, int M,
    int ksub, int dsub, uint8_t* __restrict__ codes, int imi_nbits) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][d]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + wave;
    if (v >= n) return;
    float* r = sm + wave * d;
    const int64_t key = assign ? assign[v] : -1;
    for (int c = lane; c < d; c += 64) {
        float xv = x[v * d + c];
        if (by_residual) {   // IndexIVFPQ.cpp:219-225
            if (key < 0) xv = 0.f;
            else if (imi_nbits > 0) {
                const int dc = d >> 1, m = c / dc;
                const int64_t kc = int64_t(1) << imi_nbits;
                const int64_t idx = m == 0 ? (key & (kc - 1)) : (key >> imi_nbits);
                xv = __fsub_rn(xv, coarse[((int64_t)m * kc + idx) * dc + (c - m * dc)]);
            } else xv = __fsub_rn(xv, coarse[key * d + c]);
        }
        r[c] = xv;
    }
    __builtin_amdgcn_wave_barrier();
    for (int m = 0; m < M; m++) {
        const float* xs = r + m * dsub;
        float best = 1e20f;
        int bi = -1;
        for (int j = lane; j < ksub; j += 64) {
            const float* cj = cent + ((size_t)m * ksub + j) * dsub;
            const float dis = l2sqr_sse_order([&](int c) { return xs[c]; }, [&](int c) { return cj[c]; }, dsub);
            if (dis < best) { best = dis; bi = j; }
        }
        // lexicographic (dis, idx) min across lanes; idx -1 (no candidate < 1e20) loses
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const float ob = __shfl_xor(best, sft, 64);
            const int oi = __shfl_xor(bi, sft, 64);
            const bool take = (oi >= 0) && (bi < 0 || ob < best || (ob == best && oi < bi));
            if (take) { best = ob; bi = oi; }
        }
        if (lane == 0) codes[v * M + m] = (uint8_t)bi;
    }
}

void launch_residual_encode(const float* x, int64_t n, int d, const float* coarse,
                            const int64_t* assign, int by_residual, const float* cent, int M,
                            int ksub, int dsub, uint8_t* codes, hipStream_t s, int imi_nbits) {
    if (n <= 0) return;
    const size_t smem = (size_t)4 * d * sizeof(float);
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), smem, s, x, n, d,
                       coarse, assign, by_residual, cent, M, ksub, dsub, codes, imi_nbits);
}

}  // namespace vlq
