// Coarse assignment with a float16 SCREEN in front of the exact fp32 arithmetic (round 3).
//
// The exact stage -- knn_L2sqr's (|x|^2 + |y|^2) - 2 <x, y> (utils.cpp:884) with the inner products accumulated over
// k = 0, 1, 2, ... in fp32, what the f32 MFMA distance kernel of kernels.hip computes -- spends 120 us of the 150 us
// coarse stage of the bench batch on 10 GFLOP of f32 MFMA whose results are, all but nprobe of 4096 per row, thrown
// away.  Here the whole matrix is first computed APPROXIMATELY from float16 copies of the queries and centroids
// (v_mfma_f32_32x32x16_f16: 16 x the f32 MFMA rate, so that kernel is bound by the matrix it writes), every column
// that can still belong to the row's nprobe nearest is kept -- with a rigorous bound on |approximate - exact|, below
// -- and only the kept columns (nprobe + a few) get their EXACT distance, as a k-ascending fmaf chain: bit for bit the
// value of the f32 MFMA kernel.  The selection over the kept (distance, column) keys is the exact stage's.  Same keys,
// same distances, same tie order as the matrix path; which columns were screened out never shows.
//
// Bound.  q~ = half(s q), c~ = half(s c) with s a power of two chosen from max |c_ij| (round to nearest:
// |x~ - s x| <= 2^-11 |s x| + 2^-25, the second term covers the subnormal range -- budgeted as 2^-14, what flushing
// subnormal operands to zero would cost; a query component that overflows sends its row to the exact path).  The f16
// products are exact in fp32 and the MFMA accumulates 128 of them in fp32 (error <= 2^-16 of their absolute sum,
// generously).  With |q|, |c| the Euclidean norms,
//     |ip~ / s^2 - <q, c>|  <=  (2^-10 + 2^-16 + 2^-20) |q| |c|  +  2^-14 sqrt(d) (|q| + |c|) / s
// (Cauchy-Schwarz on sum |q_i c_i| and on sum |c_i|).  The exact stage's own fp32 value differs from the real-number
// distance by at most 2^-15 (|q| + |c|)^2 (128-term fmaf chain, the two norms, three more roundings).  Hence
//     |approximate - exact|  <=  delta(q) := 1.04 * 2^-9 |q| C + 2^-13 sqrt(d) (|q| + C) / s + 2^-15 (|q| + C)^2,
// C = the largest centroid norm.  Let cut >= the nprobe-th smallest approximate distance of the row.  nprobe columns
// have exact distance <= cut + delta, so the nprobe-th smallest EXACT distance is <= cut + delta, and a column whose
// approximate distance exceeds cut + 2 delta has exact distance > cut + delta: it cannot be among the nprobe nearest,
// nor tie with the nprobe-th.  Rows where that test keeps fewer than nprobe or more than 256 columns (NaN / infinite
// input, degenerate data) are recomputed in full by the same fmaf chains.
//
// Centring.  |q - c| = |(q - mu) - (c - mu)| for every mu, and the bound is proportional to |q| |c|: the half copies, the
// norms of the approximate matrix and delta are those of q - mu and c - mu, mu = the centroids' mean (descriptor data are
// non-negative: their mean carries most of the norm).  fl(q_i - mu_i) is the real difference within 2^-24 relative --
// absorbed by the 1.04 above -- and the approximate norms |q - mu|^2, |c - mu|^2 within 2^-17, inside a 2^-15 (|q - mu| + C')^2
// term of their own.  The EXACT distances are computed from the original q, c and their norms, untouched by any of this --
// which is why the 2^-15 (|q| + C)^2 of the exact stage's own rounding keeps the UNcentred norms: data far from the origin
// have exact fp32 distances that are noise at the scale of their differences, and reproducing that noise needs every column.
// The stored matrix is half(sd * approximate distance) (the 164 MB fp32 matrix was what both kernels were bound by): its
// rounding, 2^-11 of the stored value, enters the keep test as a relative widening of both sides (coarse_screen_keep_kernel).
#include <algorithm>
#include <cmath>

#include "kernels.h"
#include "sse_order.cuh"
#include "wave_topk.cuh"

namespace vlq {

namespace {

#define FLT_MAX_F 3.402823466e+38f
__device__ __forceinline__ uint32_t f32_to_ordered_inv(uint32_t u) { return __float_as_uint(ordered_to_f32(u)); }
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// approximate distances: out[row][col] = (qn[row] + cn[col]) - 2 * <q~, c~> / s^2.  A wave owns a 64 x 64 tile and
// takes its operands straight from global memory in the MFMA's own layout (lane (r, h): row / column r of a 32-block,
// components 16 ks + 8 h .. + 7 = one 16-byte load): no LDS, no barrier.  The kernel is bound by the matrix it writes.
// tmin != nullptr (rows wider than 8192 columns): also the minimum of every 64-column tile of a row, [nq][nlist / 64];
// out == nullptr: ONLY those minima, in fp32 (the 1-NN screen of the assignment).
// (A bit-epilogue form of this kernel -- one bit per element under the row's bound instead of the matrix -- was the first
// matrix-free pass of round 5; coarse_f16_stream_kernel below replaced it, docs/EXPERIMENTS.md.)
template <int KS>
__global__ __launch_bounds__(256) void coarse_f16_dist_kernel(const _Float16* __restrict__ Qh, const _Float16* __restrict__ Ch,
                                                              const float* __restrict__ qn, const float* __restrict__ cn,
                                                              _Float16* __restrict__ out, int64_t nq, int nlist, float inv_s2,
                                                              float sd, float* __restrict__ tmin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + (wave >> 1) * 64;
    const int col0 = blockIdx.y * 128 + (wave & 1) * 64;
    if (col0 >= nlist) return;                                 // nlist % 64 == 0: a 64-column tile is whole or absent
    h16x8 a[2][KS], b[2][KS];             // blocked operand order (screen_prep_kernel): one contiguous KB per load
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const h16x8* src = reinterpret_cast<const h16x8*>(Qh) + ((row0 >> 5) + rb) * (KS * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) a[rb][ks] = src[ks * 64];
    }
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        const h16x8* src = reinterpret_cast<const h16x8*>(Ch) + (int64_t)((col0 >> 5) + cb) * (KS * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) b[cb][ks] = src[ks * 64];
    }
    float tmv[2][4];                      // tmin != nullptr: the minimum of this wave's 64 columns per row (row 8g + 4h + (r & 3) of block rb)
    float qnr[2][16];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int64_t row = row0 + rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            qnr[rb][reg] = qn[row < nq ? row : nq - 1];
        }
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
            f32x16 acc;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) acc[reg] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rb][ks], b[cb][ks], acc, 0, 0, 0);
            // lane (r, h) holds column r of rows 8g + 4h + i (register 4g + i): a 4 x 4 transpose inside each lane quad
            // (two DPP exchange steps) turns that into one row x 4 consecutive columns = one 16-byte store
            const float cnv = cn[col0 + cb * 32 + r];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    v[i] = __fsub_rn(__fadd_rn(qnr[rb][4 * g + i], cnv), __fmul_rn(2.f, __fmul_rn(acc[4 * g + i], inv_s2)));
                {
                    const bool odd = lane & 1;
                    float s0 = odd ? v[0] : v[1], s1 = odd ? v[2] : v[3];
                    s0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0xB1, 0xf, 0xf, false));
                    s1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1, 0xf, 0xf, false));
                    if (odd) { v[0] = s0; v[2] = s1; } else { v[1] = s0; v[3] = s1; }
                }
                {
                    const bool up = lane & 2;
                    float s0 = up ? v[0] : v[2], s1 = up ? v[1] : v[3];
                    s0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0x4E, 0xf, 0xf, false));
                    s1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x4E, 0xf, 0xf, false));
                    if (up) { v[0] = s0; v[1] = s1; } else { v[2] = s0; v[3] = s1; }
                }
                // (rows past nq land in the matrix's padding: the caller sizes it to whole 128-row blocks)
                // stored as half(sd * value): the matrix is what this kernel and the next are bound by.  Values beyond the
                // half range become +inf and are never kept (sd is chosen so that the range covers 4 C^2).
                const int64_t row = row0 + rb * 32 + 8 * g + 4 * h + (r & 3);
                typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                h16x4 hv;
#pragma unroll
                for (int i = 0; i < 4; i++) hv[i] = (_Float16)__fmul_rn(sd, v[i]);
                if (out) *reinterpret_cast<h16x4*>(out + row * nlist + col0 + cb * 32 + (r & ~3)) = hv;
                if (tmin) {
                    // the row's minimum over the 8 lanes that hold its other columns of this 32-block (lane bits 2..4)
                    float m = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
                    m = fminf(m, __uint_as_float(lane_xor_u32(__float_as_uint(m), 4)));
                    m = fminf(m, __uint_as_float(lane_xor_u32(__float_as_uint(m), 8)));
                    m = fminf(m, __uint_as_float(lane_xor_u32(__float_as_uint(m), 16)));
                    tmv[rb][g] = cb == 0 ? m : fminf(tmv[rb][g], m);
                }
            }
        }
    if (tmin) {
        // stored like the matrix: float(half(sd * minimum)) -- rounding is monotone, so this IS the minimum of the stored halves
        if (r < 4) {
            const int ntile = nlist >> 6;
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int64_t row = row0 + rb * 32 + 8 * g + 4 * h + r;
                    // (out == nullptr: the 1-NN screen, minima only -- kept in fp32, no matrix at all)
                    if (row < nq) tmin[row * ntile + (col0 >> 6)] = out ? (float)(_Float16)__fmul_rn(sd, tmv[rb][g]) : tmv[rb][g];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The matrix-free screen's two passes (round 5).  A wave owns 32 queries and a RANGE of 32 x NB centroids: the queries'
// operand stays in registers, the centroid blocks stream past it (next block requested while the current one multiplies);
// the four waves of a workgroup take four query blocks against the same range, so a centroid block is fetched from L2 once
// per workgroup.  Rounds 3-4's kernel gave a wave one 64 x 64 tile: 32 KB of operands per 4096 products, 640 MB of L2 reads
// at 10 000 x 4096 -- that, not the matrix it wrote, was what it was bound by (37 us with nothing written).
// The product is taken TRANSPOSED (centroids as the MFMA's rows): lane (n, h) then holds 16 centroids of ONE query n, so
//   pass 0 keeps a running minimum per (lane, register) over the range's blocks -- the row's pool for the bound: class
//          m = centroid mod 32 of every range, [nq][32 x ranges] -- 1 fma + 1 min per product, nothing across lanes;
//   pass 1 sets a bit per product under the row's bound, 16 per lane, the two halves' bits joined into the (query, block)
//          word, transposed through LDS into bits[row][col / 32].
// The approximate distance is a = qn_c + v', v' = fma(-2 / s^2, ip~, cn_c); both passes compute v' by the same operation, the
// bound kernel adds qn_c (monotone: the nprobe-th smallest class minimum of v' gives the nprobe-th smallest of a) and hands
// pass 1 the bound minus qn_c, rounded up.
#ifndef VLQ_STREAM_ABL
#define VLQ_STREAM_ABL 0          // timing experiments (wrong results): 1 no epilogue, 2 no MFMA, 4 no stage copies, 8 no LDS fragment reads
#endif
template <int KS, int PASS>
__global__ __launch_bounds__(256, 2) void coarse_f16_stream_kernel(const _Float16* __restrict__ Qh, const _Float16* __restrict__ Ch,
                                                                   const float* __restrict__ cn, int64_t nq, int nlist, int nb_range,
                                                                   float m2_inv_s2, float* __restrict__ pool, int npool,
                                                                   const float* __restrict__ tsub, uint32_t* __restrict__ bits, int nsub) {
    // The centroid blocks reach the four waves through LDS: stages of SB blocks in a ring of RING buffers, filled by LDS-DMA
    // (global_load_lds_dwordx4: a wave instruction moves 1 KB, no registers) three stages ahead -- every wave issues a quarter
    // of a stage (a linear copy: the operand order keeps a stage's blocks contiguous), waits for its own share of the stage
    // that is due (vmcnt: the two later stages stay in flight) and meets the others at ONE barrier per stage.  Measured on the
    // way here (10 000 x 4096, d = 128, per pass): every wave fetching its own blocks from L2 25-30 us whatever the prefetch
    // depth (4 x the bytes: L2 bandwidth); through LDS with register staging one stage ahead 22-27 us (a stage's loads, ~1.5 us
    // from L2 under load, against 0.7 us of multiplies).
    // A wave multiplies QW = 2 query blocks (64 queries) against every centroid block: the bytes that must be in flight to
    // cover the ~1.5 us a stage takes from L2 are halved -- with one query block per wave the loop waited for its stages
    // (22-27 us per pass in every variant of the staging).
    constexpr int SB = 2;                         // blocks per stage
    constexpr int RING = 3;
    constexpr int AHEAD = 2;
    constexpr int QW = 2;                         // query blocks per wave
    constexpr int NBR = 16;                       // blocks per range at most (the launch's nb_range)
    constexpr int PT = (SB * KS + 3) / 4;         // 1 KB pieces per wave and stage (the last may run past the stage: clamped)
    constexpr int STAGE = SB * KS * 64;           // 16-byte elements per stage
    __shared__ __attribute__((aligned(16))) h16x8 cl[RING][PT * 256];
    __shared__ uint32_t wt[PASS == 1 ? 4 * QW : 1][32][NBR + 1];  // pass 1: [wave, query block][query][block of the range] (padded)
    __shared__ __attribute__((aligned(16))) float cnl[NBR * 32];  // the range's centroid norms
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int64_t nrb = (nq + 127) / 128 * 4;                     // query blocks of 32 the half copy holds (zero rows past nq)
    const int64_t rb0 = ((int64_t)blockIdx.x * 4 + wave) * QW;    // this wave's first query block
    // (round 6) a workgroup takes nsub consecutive ranges one after the other -- indexes of more than 16 384 lists: the bound's
    // pool stays at <= 1024 class minima per row (class m = centroid mod 32 of every SUPER-range of nsub ranges), pass 0's running
    // minima simply run on over the sub-ranges, pass 1 leaves its bits range by range
    const int srange = blockIdx.y;
    const int nblk = nlist >> 5;
    // set-up loads first (older than every stage: a wait for them never waits for a stage)
    h16x8 qf[QW][KS];
#pragma unroll
    for (int qb = 0; qb < QW; qb++) {
        const h16x8* src = reinterpret_cast<const h16x8*>(Qh) + min(rb0 + qb, nrb - 1) * (KS * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[qb][ks] = src[ks * 64];
    }
    float tvu[QW];                                // pass 1's bound, one ulp up: v <= tsub  <=>  v - nextup(tsub) < 0 (the sign bit is collected)
#pragma unroll
    for (int qb = 0; qb < QW; qb++) {
        tvu[qb] = 0.f;
        if (PASS == 1) {
            const int64_t r_ = (rb0 + qb) * 32 + n;
            const float tv = tsub[r_ < nq ? r_ : nq - 1];
            tvu[qb] = __uint_as_float(f32_to_ordered_inv(f32_to_ordered(tv) + 1u));
        }
    }
    float mn[QW][16];
#pragma unroll
    for (int qb = 0; qb < QW; qb++)
#pragma unroll
        for (int r = 0; r < 16; r++) mn[qb][r] = FLT_MAX_F;
    for (int sr = 0; sr < nsub; sr++) {
    const int range = srange * nsub + sr;
    const int cb0 = range * nb_range;                             // first centroid block of 32
    if (cb0 >= nblk) break;                                       // (workgroup-uniform)
    const int nb = min(nb_range, nblk - cb0);                     // >= 1; even (nlist % 64 == 0, nb_range even)
    const h16x8* cbase = reinterpret_cast<const h16x8*>(Ch) + (int64_t)cb0 * (KS * 64);
    const int nel = nb * KS * 64;                                 // 16-byte elements of the range
    // every stage is PT instructions per wave, past the range's end too (the last element again, into a buffer nobody reads):
    // the count in flight at a wait is then a constant
    auto stage_issue = [&](int st) __attribute__((always_inline)) {
        const int buf = st % RING;
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const int e = min(st * STAGE + i * 256 + (int)threadIdx.x, nel - 1);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(cbase + e),
                                             (void __attribute__((address_space(3)))*)(&cl[buf][i * 256 + wave * 64]), 16, 0, 0);
        }
    };
    float cnr[(NBR * 32 + 255) / 256];
#pragma unroll
    for (int i = 0; i < (NBR * 32 + 255) / 256; i++) cnr[i] = cn[(int64_t)cb0 * 32 + min(i * 256 + (int)threadIdx.x, nb * 32 - 1)];
    stage_issue(0);
    stage_issue(1);
#pragma unroll
    for (int i = 0; i < (NBR * 32 + 255) / 256; i++) cnl[i * 256 + threadIdx.x] = cnr[i];
    const int nst = nb / SB;
    for (int st = 0; st < nst; st++) {
        const int buf = st % RING;
        // this wave's share of stage st has landed (stage st + 1 may still be in flight), then everybody's; the
        // barrier also says that everybody is done with stage st - 1, whose buffer the next issue refills
        __builtin_amdgcn_s_waitcnt(0x0F70 | (((AHEAD - 1) * PT) & 15) | (((((AHEAD - 1) * PT) >> 4) & 3) << 14));
        __syncthreads();
        stage_issue(st + AHEAD);
#pragma unroll
        for (int p = 0; p < SB; p++) {
            const int bb = st * SB + p;
            // (ds_read by hand: a compiler-visible LDS load behind an LDS-DMA gets a vmcnt(0) in front -- a wait for the
            // stages just requested.  What orders these reads behind their stage is the counted wait + barrier above.)
            h16x8 cf[KS];
            const uint32_t la = (uint32_t)(uintptr_t)(&cl[buf][p * KS * 64 + lane]);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cf[ks]) : "v"(la), "n"(ks * 1024));
            f32x16 acc[QW];
#pragma unroll
            for (int qb = 0; qb < QW; qb++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[qb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                // fragment ks is back when at most KS - 1 - ks of the later reads are outstanding
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cf[ks]) : "n"(KS - 1 - ks));
#pragma unroll
                for (int qb = 0; qb < QW; qb++) acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cf[ks], qf[qb][ks], acc[qb], 0, 0, 0);
            }
            const float* cp = cnl + bb * 32 + 4 * h;              // the norms of this lane's 16 centroids 8 g + 4 h + i
            float c16[16];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 c4v = *reinterpret_cast<const float4*>(cp + 8 * g);
                c16[4 * g] = c4v.x; c16[4 * g + 1] = c4v.y; c16[4 * g + 2] = c4v.z; c16[4 * g + 3] = c4v.w;
            }
#pragma unroll
            for (int qb = 0; qb < QW; qb++) {
                uint32_t w = 0;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float v = __fmaf_rn(m2_inv_s2, acc[qb][r], c16[r]);
                    if (PASS == 0) mn[qb][r] = fminf(mn[qb][r], v);                    // (a NaN never becomes a minimum)
                    else {
                        // one more bit from the top: (w << 1) | sign(v - bound); centroid 8 g + 4 h + i (r = 4 g + i) ends at
                        // bit 15 - r.  (A NaN may set its bit: a column kept for nothing, the exact stage decides.)
                        const uint32_t t = __float_as_uint(__fsub_rn(v, tvu[qb]));
                        w = __builtin_amdgcn_alignbit(w, t, 31);
                    }
                }
                if (PASS == 1) {
                    // this half's 16 bits | the other half's << 16: bit 16 h' + 15 - (4 g + i) of the word = centroid 8 g + 4 h' + i
                    const uint32_t o = lane_xor_u32(w, 32);
                    if (h == 0) wt[wave * QW + qb][n][bb] = w | (o << 16);
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the repeats issued past the range's end
    __syncthreads();
    if (PASS == 1) {
#pragma unroll
        for (int qb = 0; qb < QW; qb++) {
            const int64_t row = (rb0 + qb) * 32 + n;
            if (row >= nq) continue;
            // the range's words of 32 queries: out as contiguous bytes per query
            const int nw = nlist >> 5;
            uint32_t* dst = bits + row * nw + cb0 + (NBR / 2) * h;
#pragma unroll
            for (int j = 0; j < NBR / 2; j++)
                if ((NBR / 2) * h + j < nb) dst[j] = wt[wave * QW + qb][n][(NBR / 2) * h + j];
        }
    }
    if (sr + 1 < nsub) __syncthreads();           // the next sub-range refills the norms, the ring and the word table
    }   // sub-ranges
    if (PASS == 0) {
#pragma unroll
        for (int qb = 0; qb < QW; qb++) {
            const int64_t row = (rb0 + qb) * 32 + n;
            if (row >= nq) continue;
            float* dst = pool + row * npool + srange * 32 + 4 * h;    // class m = 8 g + 4 h + i of this super-range
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4*>(dst + 8 * g) = make_float4(mn[qb][4 * g], mn[qb][4 * g + 1], mn[qb][4 * g + 2], mn[qb][4 * g + 3]);
        }
    }
}

// exact distance of (query row staged in LDS, centroid col): the fmaf chain of the f32 MFMA kernel over k = 0, 1, 2, ...,
// utils.cpp:884's formula.  The centroid row is fetched 64 components at a time (16 outstanding 16-byte loads per lane:
// every lane reads a different row, and one load per chain step would pay its latency 32 times).
__device__ __forceinline__ float exact_distance(const float* qrow, const float* __restrict__ cj, int d, float qnv, float cnv) {
    float ip = 0.f;
    int k = 0;
    for (; k + 64 <= d; k += 64) {
        float4 cv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) cv[u] = *reinterpret_cast<const float4*>(cj + k + 4 * u);
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const float4 qv = *reinterpret_cast<const float4*>(qrow + k + 4 * u);
            ip = __fmaf_rn(qv.x, cv[u].x, ip);
            ip = __fmaf_rn(qv.y, cv[u].y, ip);
            ip = __fmaf_rn(qv.z, cv[u].z, ip);
            ip = __fmaf_rn(qv.w, cv[u].w, ip);
        }
    }
    for (; k + 4 <= d; k += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(qrow + k);
        const float4 cv = *reinterpret_cast<const float4*>(cj + k);
        ip = __fmaf_rn(qv.x, cv.x, ip);
        ip = __fmaf_rn(qv.y, cv.y, ip);
        ip = __fmaf_rn(qv.z, cv.z, ip);
        ip = __fmaf_rn(qv.w, cv.w, ip);
    }
    for (; k < d; k++) ip = __fmaf_rn(qrow[k], cj[k], ip);
    return __fsub_rn(__fadd_rn(qnv, cnv), __fmul_rn(2.f, ip));
}

// Everything the screen needs of a batch of rows x [n][d], one pass: a workgroup stages 32 rows in LDS with coalesced
// loads, then
//   * all threads write half(scale * (x - mu)) in the MFMA's operand order: rows in blocks of 32, components in steps of
//     16; block (row / 32, ks) is 1 KB = lane (r = row % 32, h = (k % 16) / 8) x 8 halves, so a wave's operand load of one
//     block is one contiguous KB.  Rows n .. n_pad - 1 (n_pad = n rounded up to 128) and components d .. 16 ks_n - 1 are zero;
//   * four threads per row, one per lane accumulator of fvec_norm_L2sqr (utils.cpp:538-556: s_l += x[4i+l]^2 in increasing i,
//     then (s0 + s1) + (s2 + s3) -- the matrix path's norm_sse_order, bit for bit: the exact distances need it), which also
//     sum their quarter of the centred row's squared norm (the approximate matrix and delta need it; any order) and of the
//     half-range flag of the centred, scaled row (flags optional).  (One thread per row until round 5: 64 of the workgroup's
//     256 threads walked 128 components twice while the others waited -- 13.2 us per 10 000 rows.)
// d % 4 == 0 (coarse_screen_shape_ok); rows padded to d + 4 floats: a quad's four lanes read consecutive words, eight rows
// cover the 32 banks.
constexpr int kPrepRows = 32;
__global__ __launch_bounds__(128) void screen_prep_kernel(const float* __restrict__ x, const float* __restrict__ mu, int64_t n, int d,
                                                          int ks_n, float scale, _Float16* __restrict__ out, float* __restrict__ norms,
                                                          float* __restrict__ norms_c, unsigned char* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) float rows[];   // [kPrepRows][d + 4]
    const int ld = d + 4;
    const int64_t row0 = (int64_t)blockIdx.x * kPrepRows;
    const int nr = (int)max((int64_t)0, min((int64_t)kPrepRows, n - row0));
    const int t = threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const float4* src = reinterpret_cast<const float4*>(x + row0 * d);
        const int d4 = d >> 2;
        for (int e = t; e < nr * d4; e += 128) *reinterpret_cast<float4*>(rows + (e / d4) * ld + 4 * (e % d4)) = src[e];
    } else {
        for (int e = t; e < nr * d; e += 128) rows[(e / d) * ld + e % d] = x[row0 * d + e];
    }
    __syncthreads();
    const int gpr = 2 * ks_n;                                  // 8-component groups per row
    // lanes along the rows: the 32 rows' pieces of one (ks, h) are 512 contiguous bytes of the operand block (lanes along the
    // groups wrote 16 bytes every KB: 64 lines per store instruction)
    for (int e = t; e < kPrepRows * gpr; e += 128) {
        const int r = e & (kPrepRows - 1), grp = e / kPrepRows, ks = grp >> 1, h = grp & 1;
        float xv[8];
        if (8 * grp + 8 <= d) {
            const float4 a = *reinterpret_cast<const float4*>(rows + r * ld + 8 * grp), b = *reinterpret_cast<const float4*>(rows + r * ld + 8 * grp + 4);
            xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w; xv[4] = b.x; xv[5] = b.y; xv[6] = b.z; xv[7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) xv[i] = 8 * grp + i < d ? rows[r * ld + 8 * grp + i] : 0.f;
        }
        h16x8 v;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int k = 8 * grp + i;
            v[i] = (_Float16)((r < nr && k < d) ? __fmul_rn(scale, __fsub_rn(xv[i], mu[k])) : 0.f);
        }
        const int64_t row = row0 + r;
        *reinterpret_cast<h16x8*>(out + (((row >> 5) * ks_n + ks) * 64 + h * 32 + (row & 31)) * 8) = v;
    }
    const int r = t >> 2, l = t & 3;
    const float* xr = rows + min(r, max(nr - 1, 0)) * ld;      // (threads of missing rows walk the last one: shuffles stay uniform)
    float sl = 0.f, nc = 0.f;
    bool bad = false;
    if (nr > 0) {
        for (int c = l; c < d; c += 4) {
            const float v = xr[c];
            sl = __fadd_rn(sl, __fmul_rn(v, v));
            const float w = __fsub_rn(v, mu[c]);
            nc = __fmaf_rn(w, w, nc);
            bad = bad || !(fabsf(__fmul_rn(scale, w)) <= 65504.f);      // also true for NaN
        }
        sl = __fadd_rn(sl, 0.f);                               // (the reference's unconditional tail add)
    }
    // (s0 + s1) + (s2 + s3): the quad's pairs, then the two pair sums
    const float pr = __fadd_rn(sl, __shfl_xor(sl, 1));
    const float nrm = __fadd_rn(pr, __shfl_xor(pr, 2));
    nc += __shfl_xor(nc, 1);
    nc += __shfl_xor(nc, 2);
    const int badq = (int)bad | __shfl_xor((int)bad, 1);
    const int bada = badq | __shfl_xor(badq, 2);
    if (l == 0 && r < nr) {
        norms[row0 + r] = nrm;
        norms_c[row0 + r] = nc;
        if (flags) flags[row0 + r] = bada ? 1 : 0;
    }
}

constexpr int kKeepCap = 512;                   // kept columns per row (typically nprobe + 20); more -> the row is done exactly

// cut: an upper bound of the row's nprobe-th smallest stored value -- the nprobe-th smallest of the 64 MT values "MT smallest
// elements of every lane" (distinct columns; one minimum per lane is too loose a pool once nprobe approaches 64: its
// nprobe-th smallest is then the LARGEST lane minimum).  MT = 2 for nprobe <= 64, 4 for nprobe <= 128.
template <int MT>
__device__ __forceinline__ void lane_top(float (&m)[MT], float x) {
#pragma unroll
    for (int i = 0; i < MT; i++) {
        const float t = fmaxf(m[i], x);
        m[i] = fminf(m[i], x);
        x = t;
    }
}
template <int MT>
__device__ __forceinline__ float screen_cut(const float (&m)[MT], int nprobe, int lane) {
    u64 p[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) p[i] = ((u64)f32_to_ordered(m[i]) << 32) | (uint32_t)(64 * i + lane);
    wave_sort_multi<MT>(p, lane);
    const int e = nprobe - 1;
    u64 row = p[0];
#pragma unroll
    for (int i = 1; i < MT; i++) row = (e >> 6) == i ? p[i] : row;
    return ordered_to_f32((uint32_t)(bcast_u64(row, e & 63) >> 32));
}

// smallest of a float over the wave, as bits (DPP inside rows of 16, readlane across; NaN inputs lose to numbers: fminf)
__device__ __forceinline__ uint32_t wave_min_ordered(float v) {
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false)));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return __float_as_uint(fminf(fminf(a, b), fminf(c, e)));
}

// T: the largest stored value (half(sd * approximate distance), read back as float) a column of this row may have and still
// be kept, from the row's cut (nprobe-th smallest stored lane minimum); *finite = the bound is a number
// tmax: the largest stored value that is still a number -- 65504 where the stored values are halves.  A bound at or above
// it could admit columns whose stored value overflowed to +inf (sd * distance >= 65520: queries far outside the centroid
// cloud, for which sd was not chosen), and `w <= T` never holds for +inf: such a row is not decided here (*finite = false ->
// the exact path).
__device__ __forceinline__ float screen_threshold(float cut_s, float qnv, float qn0v, float cmax, float cmax0, float c_sub, float inv_sd,
                                                  bool* finite, float tmax = FLT_MAX_F) {
    // delta(q), inflated: every factor rounded up generously (the bound is what exactness rests on)
    // (qn / cmax: centred -- the half arithmetic's error; qn0 / cmax0: as given -- the exact stage's own fp32 error)
    const float qnorm = __fmul_rn(sqrtf(fmaxf(qnv, 0.f)), 1.0001f);
    const float sum = __fadd_rn(qnorm, cmax);
    const float sum0 = __fadd_rn(__fmul_rn(sqrtf(fmaxf(qn0v, 0.f)), 1.0001f), cmax0);
    const float delta = __fmul_rn(1.001f, __fadd_rn(__fadd_rn(__fmul_rn(__fmul_rn(0.00203125f /* 1.04 * 2^-9 */, qnorm), cmax),
                                                              __fmul_rn(c_sub, sum)),
                                                    __fadd_rn(__fmul_rn(3.0517578125e-05f /* 2^-15 */, __fmul_rn(sum, sum)),
                                                              __fmul_rn(3.0517578125e-05f /* 2^-15 */, __fmul_rn(sum0, sum0)))));
    // a stored value w stands for an approximate distance within 2^-11 |w| / sd (+ the subnormal step) of x = w / sd: a column
    // is kept iff the smallest distance its stored value allows is <= the largest the cut's allows + 2 delta, i.e. iff
    // x - eps |x| - sub <= thr.  For thr + sub >= 0 that is x <= (thr + sub) / (1 - eps) (every negative x passes), otherwise
    // x <= (thr + sub) / (1 + eps): ONE compare per element against T (rounded up), in the stored domain.
    const float cut = __fmul_rn(inv_sd, cut_s);
    const float sub = __fmul_rn(inv_sd, 6.0e-8f /* > 2^-24, the subnormal half step */);
    const float thr = __fadd_rn(__fadd_rn(__fadd_rn(cut, __fmul_rn(0.000489f /* > 2^-11 */, fabsf(cut))), sub), __fmul_rn(2.0002f, delta));
    const float ts = __fadd_rn(thr, sub);
    const float tx = ts >= 0.f ? __fmul_rn(ts, 1.00049f /* > 1 / (1 - eps) */) : __fmul_rn(ts, 0.99951f /* < 1 / (1 + eps): towards 0 */);
    // back to the stored domain, rounded up (1 / inv_sd is the power of two sd: exact)
    const float T = __fmul_rn(__fadd_rn(tx, __fmul_rn(1e-6f, fabsf(tx))), 1.f / inv_sd);
    *finite = thr < FLT_MAX_F && T < tmax;
    return T;
}

// one wave per row: the row of approximate distances in registers, cut = nprobe-th smallest of the lanes' two smallest
// elements each, columns at or below cut + 2 delta(q) are kept: keep[q][0 .. nkeep[q]) (nkeep = 0xffff: the whole row exactly)
template <int NV, int MT>
__global__ __launch_bounds__(256) void coarse_screen_keep_kernel(const _Float16* __restrict__ dist, int64_t nq, int nlist, int nprobe,
                                                                 const float* __restrict__ qn, const float* __restrict__ qn0,
                                                                 const unsigned char* __restrict__ flags, float cmax, float cmax0,
                                                                 float c_sub, float inv_sd, uint32_t* __restrict__ keep,
                                                                 uint16_t* __restrict__ nkeep, unsigned int* __restrict__ exact_rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;                        // whole wave; no workgroup barrier below
    const float qnv = qn[q];
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    const h16x4* row4 = reinterpret_cast<const h16x4*>(dist + q * nlist);
    const int n4 = nlist >> 2;
    float4 v[NV];                               // the stored halves, still scaled by sd
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const h16x4 hv = row4[min(u * 64 + lane, n4 - 1)];      // clamped; masked below
        v[u] = make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
    }
    float mt[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) mt[i] = FLT_MAX_F;
#pragma unroll
    for (int u = 0; u < NV; u++) {
        if (u * 64 + lane >= n4) v[u] = make_float4(FLT_MAX_F, FLT_MAX_F, FLT_MAX_F, FLT_MAX_F);
        lane_top<MT>(mt, v[u].x); lane_top<MT>(mt, v[u].y); lane_top<MT>(mt, v[u].z); lane_top<MT>(mt, v[u].w);
    }
    const float cut_s = screen_cut<MT>(mt, nprobe, lane);      // scaled by sd, rounded to half
    bool finite;
    const float T = screen_threshold(cut_s, qnv, qn0[q], cmax, cmax0, c_sub, inv_sd, &finite, 65504.f);
    const bool undecided = !finite || flags[q];       // NaN / infinite / half-overflowing bound, or a query outside the half range
    // kept columns: a ballot per register component (most are empty: ~nprobe + 20 of the row's elements pass)
    uint32_t* out = keep + q * kKeepCap;
    int total = 0;
    auto take = [&](float w, uint32_t col) __attribute__((always_inline)) {
        const bool p = w <= T;                               // false for NaN and for +inf (T is finite)
        const u64 m = __ballot(p);
        if (m != 0) {
            const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
            if (p && pos < kKeepCap) out[pos] = col;
            total += __popcll(m);
        }
    };
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const uint32_t c0 = (uint32_t)(4 * (u * 64 + lane));
        take(v[u].x, c0 + 0);
        take(v[u].y, c0 + 1);
        take(v[u].z, c0 + 2);
        take(v[u].w, c0 + 3);
    }
    const bool exact_row = undecided || total > kKeepCap || total < nprobe;
    if (lane == 0) {
        nkeep[q] = exact_row ? (uint16_t)0xffff : (uint16_t)total;
        if (exact_row && exact_rows) atomicAdd(exact_rows, 1u);
    }
}

// rows wider than 8192 columns with tile minima from the distance kernel (one stored value per row and 64-column tile):
// the cut comes from the tile minima -- every tile minimum is a distinct column's value, so the nprobe-th smallest of the
// lanes' two smallest tile minima bounds the nprobe-th smallest element --, and only the tiles whose minimum passes the
// threshold are read: ~nprobe + a few tiles of 128 bytes instead of the whole row (32 KB at 16 384 columns, 256 KB at 2^17).
// Passing tiles are listed in LDS; 4 lanes then share a tile (16 halves each), 16 tiles per step.
template <int MT>
__global__ __launch_bounds__(256) void coarse_screen_keep_tiled_kernel(const _Float16* __restrict__ dist, const float* __restrict__ tmin,
                                                                       int64_t nq, int nlist, int nprobe, const float* __restrict__ qn,
                                                                       const float* __restrict__ qn0, const unsigned char* __restrict__ flags,
                                                                       float cmax, float cmax0, float c_sub, float inv_sd,
                                                                       uint32_t* __restrict__ keep, uint16_t* __restrict__ nkeep,
                                                                       unsigned int* __restrict__ exact_rows) {
    constexpr int TCAP = 256;                   // passing tiles per row; more -> the row is done exactly
    __shared__ uint32_t tl[4][TCAP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;                        // whole wave; no workgroup barrier below
    const int ntile = nlist >> 6;
    const float* tm = tmin + q * ntile;
    float mt[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) mt[i] = FLT_MAX_F;
    for (int t0 = 0; t0 < ntile; t0 += 64) {
        const int t = t0 + lane;
        if (t < ntile) lane_top<MT>(mt, tm[t]);
    }
    const float cut_s = screen_cut<MT>(mt, nprobe, lane);
    bool finite;
    const float T = screen_threshold(cut_s, qn[q], qn0[q], cmax, cmax0, c_sub, inv_sd, &finite, 65504.f);
    const bool undecided = !finite || flags[q];
    int npass = 0;
    for (int t0 = 0; t0 < ntile; t0 += 64) {
        const int t = t0 + lane;
        const bool p = t < ntile && tm[t] <= T;
        const u64 m = __ballot(p);
        if (m != 0) {
            const int pos = npass + __popcll(m & ((1ull << lane) - 1ull));
            if (p && pos < TCAP) tl[wave][pos] = (uint32_t)t;
            npass += __popcll(m);
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t* out = keep + q * kKeepCap;
    int total = 0;
    if (!undecided && npass <= TCAP) {
        for (int s0 = 0; s0 < npass; s0 += 16) {
            const int ti = s0 + (lane >> 2);
            const bool in = ti < npass;
            const uint32_t tile = in ? tl[wave][ti] : tl[wave][0];
            // 16 halves of the tile: columns tile * 64 + 16 (lane & 3) + i
            const h16x8* src = reinterpret_cast<const h16x8*>(dist + q * nlist + (int64_t)tile * 64 + 16 * (lane & 3));
            const h16x8 v0 = src[0], v1 = src[1];
            const uint32_t c0 = tile * 64u + 16u * (uint32_t)(lane & 3);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float w = (float)(i < 8 ? v0[i] : v1[i - 8]);
                const bool p = in && w <= T;
                const u64 m = __ballot(p);
                if (m != 0) {
                    const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
                    if (p && pos < kKeepCap) out[pos] = c0 + (uint32_t)i;
                    total += __popcll(m);
                }
            }
        }
    }
    const bool exact_row = undecided || npass > TCAP || total > kKeepCap || total < nprobe;
    if (lane == 0) {
        nkeep[q] = exact_row ? (uint16_t)0xffff : (uint16_t)total;
        if (exact_row && exact_rows) atomicAdd(exact_rows, 1u);
    }
}

// The matrix-free screen's bound (round 5): one wave per row reads the row's 64-column tile minima (fp32, from the minima pass:
// every tile minimum is a distinct column's value, so the nprobe-th smallest of them bounds the row's nprobe-th smallest element
// from above) and leaves trow[row] = the largest approximate distance a column may have and still be kept -- -FLT_MAX for a row
// the bound cannot decide (rflag[row] = 1: the exact kernel does the whole row).  MT = tile minima per lane (ntile <= 64 MT).
template <int MT>
__global__ __launch_bounds__(256) void coarse_screen_cut_kernel(const float* __restrict__ pool, int64_t nq, int npool, int nprobe,
                                                                const float* __restrict__ qn, const float* __restrict__ qn0,
                                                                const unsigned char* __restrict__ flags, float cmax, float cmax0, float c_sub,
                                                                float* __restrict__ tsub, unsigned char* __restrict__ rflag) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const float* pq = pool + q * npool;
    float mt[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) { const int t = i * 64 + lane; mt[i] = t < npool ? pq[t] : FLT_MAX_F; }
    // nprobe-th smallest class minimum of v' (distinct columns): only the VALUE is needed -- 32-bit sorts and merges of the
    // ordered images (nprobe <= 64: the answer is among the 64 smallest)
    // (more than two values per lane: the lane's two smallest stand for them -- still distinct columns' values, and the sorting
    // network, which is what this kernel's time is, stays at two registers)
    constexpr int SK = MT >= 2 ? 2 : 1;
    uint32_t sk[SK];
    {
        uint32_t a = f32_to_ordered(mt[0]), b2 = MT >= 2 ? f32_to_ordered(mt[MT >= 2 ? 1 : 0]) : 0xffffffffu;
        if (MT >= 2) { const uint32_t lo = min(a, b2), hi = max(a, b2); a = lo; b2 = hi; }
#pragma unroll
        for (int i = 2; i < MT; i++) {
            const uint32_t x = f32_to_ordered(mt[i]);
            const uint32_t t = max(a, x);
            a = min(a, x);
            b2 = min(b2, t);
        }
        sk[0] = wave_sort64_u32(a, lane);
        if (MT >= 2) sk[SK - 1] = wave_sort64_u32(b2, lane);
    }
    // the 64 smallest of two ascending lists: elementwise min against the other list reversed (a bitonic sequence), re-sorted
    auto lo64 = [&](uint32_t x, uint32_t o) __attribute__((always_inline)) {
        o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x140, 0xF, 0xF, true);      // row_mirror
        o = lane_xor_u32(lane_xor_u32(o, 16), 32);
        uint32_t v = min(x, o);
        { const uint32_t t = lane_xor_u32(v, 32); v = (lane & 32) == 0 ? min(v, t) : max(v, t); }
        { const uint32_t t = lane_xor_u32(v, 16); v = (lane & 16) == 0 ? min(v, t) : max(v, t); }
        { const uint32_t t = lane_xor_u32(v, 8); v = (lane & 8) == 0 ? min(v, t) : max(v, t); }
        { const uint32_t t = lane_xor_u32(v, 4); v = (lane & 4) == 0 ? min(v, t) : max(v, t); }
        { const uint32_t t = lane_xor_u32(v, 2); v = (lane & 2) == 0 ? min(v, t) : max(v, t); }
        { const uint32_t t = lane_xor_u32(v, 1); v = (lane & 1) == 0 ? min(v, t) : max(v, t); }
        return v;
    };
    if (MT >= 2) sk[0] = lo64(sk[0], sk[SK - 1]);
    float cutv = ordered_to_f32((uint32_t)__builtin_amdgcn_readlane((int)sk[0], min(nprobe, 64) - 1));
    if (nprobe > 64) {
        // (round 6: 64 < nprobe <= 128) the nprobe-th smallest of ALL the pool's values, exactly: bisection on the ordered image --
        // the smallest K with at least nprobe values <= K (32 steps of MT compares per lane and one count over the wave)
        uint32_t lo = 0u, hi = 0xFFFFFFFFu;
        uint32_t om[MT];
#pragma unroll
        for (int i = 0; i < MT; i++) om[i] = f32_to_ordered(mt[i]);
        for (int it = 0; it < 32; it++) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            int c = 0;
#pragma unroll
            for (int i = 0; i < MT; i++) c += om[i] <= mid ? 1 : 0;
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) c += (int)lane_xor_u32((uint32_t)c, sft);
            if (c >= nprobe) hi = mid; else lo = mid + 1u;
        }
        cutv = ordered_to_f32(lo);
    }
    const float qc = qn[q];
    const float cut = __fadd_rn(qc, cutv);                        // ... of the approximate distances a = qn_c + v' (monotone)
    bool finite;
    const float T = screen_threshold(cut, qc, qn0[q], cmax, cmax0, c_sub, 1.f, &finite);      // (fp32 values: sd = 1)
    const bool undecided = !finite || flags[q] || !(cutv < FLT_MAX_F);
    // v' > tsub  =>  fl(qn_c + v') > T: the difference rounded up by more than the two roundings can lose
    const float ts = __fadd_rn(__fsub_rn(T, qc), __fmul_rn(4.8e-7f /* 2^-21 */, __fadd_rn(fabsf(T), fabsf(qc))));
    if (lane == 0) { tsub[q] = undecided ? -FLT_MAX_F : ts; rflag[q] = undecided ? 1 : 0; }
}

// <q, c_col> for up to 64 columns at once, lane l for column col_of(l) (l < ncand), as the f32 MFMA kernel accumulates it: an
// fmaf chain over k = 0, 1, 2, ...  The centroid rows come in through LDS, 16 components at a time: 4 lanes fetch one row's
// 64-byte piece (a lane reading its own row would touch 64 cache lines per load instruction), the owner reads its row back
// (rows padded to 20 floats: 16-byte LDS accesses both ways).  qrow: the query in LDS; st: 64 x 20 floats of LDS per wave.
template <typename ColOf>
__device__ __forceinline__ float exact_ip_batch(const float* __restrict__ Cn, int d, ColOf col_of, int ncand, const float* qrow, float* st,
                                                int lane) {
    const int nch = (d + 15) >> 4;
    // pieces of 4 chunks (64 components) are in flight at a time: a load's latency here is ~2 us, the chain
    // of one chunk takes a tenth of that
    float4 piece[4][4];
    // the four rows this lane helps to fetch (lanes 4a .. 4a+3: row a + 16 j), their pieces of a chunk 64 bytes apart
    const float* rp[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int ci = (lane >> 2) + 16 * j;
        rp[j] = Cn + (size_t)(ci < ncand ? col_of(ci) : col_of(0)) * d + 4 * (lane & 3);
    }
    auto fetch = [&](int ch, float4 (&dst)[4]) __attribute__((always_inline)) {
        const bool ld = 16 * ch + 4 * (lane & 3) < d;        // (rows past ncand read row 0 of the batch: never used)
#pragma unroll
        for (int j = 0; j < 4; j++)
            dst[j] = ld ? *reinterpret_cast<const float4*>(rp[j] + 16 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float ip = 0.f;
    for (int ch0 = 0; ch0 < nch; ch0 += 4) {
#pragma unroll
        for (int c4 = 0; c4 < 4; c4++) fetch(ch0 + c4, piece[c4]);      // (chunks past d load nothing)
#pragma unroll
        for (int c4 = 0; c4 < 4; c4++) {
            const int ch = ch0 + c4;
            if (ch < nch) {                    // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<float4*>(st + ((lane >> 2) + 16 * j) * 20 + 4 * (lane & 3)) = piece[c4][j];
                __builtin_amdgcn_wave_barrier();
                const float* mine = st + lane * 20;
                const int k0 = 16 * ch, kn = min(16, d - k0);
                for (int kk = 0; kk < kn; kk += 4) {
                    const float4 qv = *reinterpret_cast<const float4*>(qrow + k0 + kk);
                    const float4 cv = *reinterpret_cast<const float4*>(mine + kk);
                    ip = __fmaf_rn(qv.x, cv.x, ip);
                    ip = __fmaf_rn(qv.y, cv.y, ip);
                    ip = __fmaf_rn(qv.z, cv.z, ip);
                    ip = __fmaf_rn(qv.w, cv.w, ip);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    return ip;
}

// one wave per row: exact distances of the kept columns -- a lane owns one column and runs the f32 MFMA kernel's fmaf
// chain over k = 0, 1, 2, ... (utils.cpp:884's formula around it) -- and the exact (distance, column) selection.  The
// centroid rows come in through LDS, 16 components at a time: 4 lanes fetch one row's 64-byte piece (a lane reading
// its own row would touch 64 cache lines per load instruction), the owner reads its row back component by component
// (rows padded to 20 floats: 16-byte LDS accesses both ways); the next chunk's pieces are requested before the current one is used.
// BITS: the kept columns come as the row's bitmap (bits[q][nlist / 32], the matrix-free screen) with rflag[q] = "whole row";
// `keep` / `nkeep` are then unused.
template <int KPL, bool BITS = false>
__global__ __launch_bounds__(256) void coarse_screen_exact_kernel(const uint32_t* __restrict__ keep, const uint16_t* __restrict__ nkeep,
                                                                  int64_t nq, int nlist, int nprobe, float* __restrict__ cdis,
                                                                  int64_t* __restrict__ keys, const float* __restrict__ Q,
                                                                  const float* __restrict__ Cn, const float* __restrict__ qn,
                                                                  const float* __restrict__ cn, int d,
                                                                  unsigned long long* __restrict__ kept_total,
                                                                  const uint32_t* __restrict__ bits = nullptr,
                                                                  const unsigned char* __restrict__ rflag = nullptr,
                                                                  unsigned int* __restrict__ exact_rows = nullptr, OrderHist oh = OrderHist()) {
    __shared__ u64 queue[4][64];
    __shared__ uint32_t cand[4][kKeepCap];
    __shared__ __attribute__((aligned(16))) float qrow[4][128];
    __shared__ __attribute__((aligned(16))) float stage[4][64 * 20];      // rows padded to 20 floats: 16-byte accesses both ways
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;                        // whole wave; no workgroup barrier below
    // one round trip for everything the row needs first: the count, the whole (fixed-size) list, the query, its norm
    const float q0 = lane < d ? Q[q * d + lane] : 0.f, q1 = lane + 64 < d ? Q[q * d + lane + 64] : 0.f;
    const float qnv = qn[q];
    bool exact_row;
    int total;
    if (BITS) {
        // the row's bitmap -> column list in LDS (ascending columns): 64 words per step, positions by a prefix sum of the popcounts
        const int nw = nlist >> 5;
        const uint32_t* bq = bits + q * nw;
        const bool whole = rflag[q] != 0;
        int tot = 0;
        for (int w0 = 0; w0 < nw; w0 += 64) {
            const int w = w0 + lane;
            uint32_t word = (w < nw && !whole) ? bq[w] : 0u;
            const uint32_t c = (uint32_t)__popc(word);
            const uint32_t incl = wave_scan_incl_u32(c);
            uint32_t pos = (uint32_t)tot + incl - c;
            while (word) {
                const int bpos = __ffs(word) - 1;
                // (the stream kernel's bit order: bit 16 h + 15 - (4 g + i) is column 8 g + 4 h + i of the word's 32)
                const uint32_t k15 = 15u - ((uint32_t)bpos & 15u), hh = (uint32_t)bpos >> 4;
                if (pos < (uint32_t)kKeepCap) cand[wave][pos] = (uint32_t)w * 32u + 8u * (k15 >> 2) + 4u * hh + (k15 & 3u);
                pos++;
                word &= word - 1u;
            }
            tot += (int)__builtin_amdgcn_readlane((int)incl, 63);
        }
        exact_row = whole || tot > kKeepCap || tot < nprobe;
        total = exact_row ? 0 : tot;
        if (exact_row && exact_rows && lane == 0) atomicAdd(exact_rows, 1u);
    } else {
        const int nk = nkeep[q];
        uint32_t kc[kKeepCap / 64];
#pragma unroll
        for (int u = 0; u < kKeepCap / 64; u++) kc[u] = keep[q * kKeepCap + u * 64 + lane];
        exact_row = nk == 0xffff;
        total = exact_row ? 0 : nk;
#pragma unroll
        for (int u = 0; u < kKeepCap / 64; u++) cand[wave][u * 64 + lane] = kc[u];
    }
    qrow[wave][lane] = q0;
    qrow[wave][lane + 64] = q1;
    WaveSelect<KPL> sel;
    sel.init(nprobe, queue[wave], lane);
    __builtin_amdgcn_wave_barrier();
#ifdef VLQ_EXACT_PROLOGUE_ONLY
    if (total >= 0) { if (lane == 0) cdis[q * nprobe] = (float)total; return; }
#endif
    if (!exact_row) {
        float* st = stage[wave];
        for (int c0 = 0; c0 < total; c0 += 64) {
            const int ncand = min(64, total - c0);
            const bool valid = lane < ncand;
            const uint32_t col = valid ? cand[wave][c0 + lane] : 0u;
            const float cnv = cn[col];             // (in flight with the pieces)
            const float ip = exact_ip_batch(Cn, d, [&](int ci) { return cand[wave][c0 + ci]; }, ncand, qrow[wave], st, lane);
            const float x = __fsub_rn(__fadd_rn(qnv, cnv), __fmul_rn(2.f, ip));
            // columns do not arrive in increasing order: equal distances are queued, the key decides
            sel.template offer<false>(x, col, valid);
        }
        if (kept_total && lane == 0) atomicAdd(kept_total, (unsigned long long)total);
    } else {
        // the whole row exactly, one column per lane, ascending columns (the ordered rule is exact)
        for (int j0 = 0; j0 < nlist; j0 += 64) {
            const int j = j0 + lane;
            const bool valid = j < nlist;
            const int jc = valid ? j : 0;
            const float x = exact_distance(qrow[wave], Cn + (size_t)jc * d, d, qnv, cn[jc]);
            sel.offer(x, (uint32_t)j, valid);
        }
        if (kept_total && lane == 0) atomicAdd(kept_total, (unsigned long long)nlist);
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
            if (e == 0 && oh.hist) {            // the scan order's histogram over the rows' nearest centroids (qorder_hist_kernel's bins)
                const int64_t k0 = miss ? -1 : (int64_t)(uint32_t)key;
                const bool ok = k0 >= 0 && k0 < oh.nlist;
                atomicAdd(&oh.hist[ok ? ((oh.list_rank ? oh.list_rank[k0] : (int)k0) >> oh.shift) : oh.nbins - 1], 1);
            }
        }
    }
}

// 1-NN (the assignment of add / encode; nprobe == 1): no matrix.  One wave per row: the smallest approximate tile minimum
// is the cut; a tile whose approximate minimum exceeds cut + 2 delta holds no column that could be the nearest or tie with
// it (its exact distances all exceed cut + delta >= the nearest's); the 64 columns of every other tile -- one or two
// tiles, usually -- get their exact distances, and the row's answer is the smallest (distance, column) key, the matrix
// path's arg-min.  More than 8 tiles under the bound, or a bound that is not a number: the whole row exactly.
__global__ __launch_bounds__(256) void coarse_screen_nn_kernel(const float* __restrict__ tmin, int64_t nq, int nlist, float* __restrict__ cdis,
                                                               int64_t* __restrict__ keys, const float* __restrict__ Q,
                                                               const float* __restrict__ Cn, const float* __restrict__ qn,
                                                               const float* __restrict__ cn, const float* __restrict__ qn_c,
                                                               const unsigned char* __restrict__ flags, int d, float cmax, float cmax0,
                                                               float c_sub, unsigned int* __restrict__ exact_rows) {
    constexpr int TCAP = 8;
    __shared__ uint32_t tl[4][TCAP];
    __shared__ __attribute__((aligned(16))) float qrow[4][128];
    __shared__ __attribute__((aligned(16))) float stage[4][64 * 20];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;                        // whole wave; no workgroup barrier below
    const int ntile = nlist >> 6;
    const float* tm = tmin + q * ntile;
    const float q0 = lane < d ? Q[q * d + lane] : 0.f, q1 = lane + 64 < d ? Q[q * d + lane + 64] : 0.f;
    const float qnv = qn[q];
    float mn = FLT_MAX_F;
    for (int t = lane; t < ntile; t += 64) mn = fminf(mn, tm[t]);
    const float cut = __uint_as_float(wave_min_ordered(mn));
    bool finite;
    const float T = screen_threshold(cut, qn_c[q], qnv, cmax, cmax0, c_sub, 1.f, &finite);      // (fp32 minima: sd = 1)
    qrow[wave][lane] = q0;
    qrow[wave][lane + 64] = q1;
    int npass = 0;
    for (int t0 = 0; t0 < ntile; t0 += 64) {
        const int t = t0 + lane;
        const bool p = t < ntile && tm[t] <= T;
        const u64 m = __ballot(p);
        if (m != 0) {
            const int pos = npass + __popcll(m & ((1ull << lane) - 1ull));
            if (p && pos < TCAP) tl[wave][pos] = (uint32_t)t;
            npass += __popcll(m);
        }
    }
    __builtin_amdgcn_wave_barrier();
    const bool exact_row = !finite || flags[q] || npass > TCAP || npass < 1;
    u64 best = kMaxKey;
    if (!exact_row) {
        for (int i = 0; i < npass; i++) {
            const uint32_t col = tl[wave][i] * 64u + (uint32_t)lane;
            const float cnv = cn[col];
            const float ip = exact_ip_batch(Cn, d, [&](int ci) { return tl[wave][i] * 64u + (uint32_t)ci; }, 64, qrow[wave], stage[wave], lane);
            best = umin64(best, make_key(__fsub_rn(__fadd_rn(qnv, cnv), __fmul_rn(2.f, ip)), col));
        }
    } else {
        for (int j0 = 0; j0 < nlist; j0 += 64) {
            const int j = j0 + lane;
            if (j < nlist) best = umin64(best, make_key(exact_distance(qrow[wave], Cn + (size_t)j * d, d, qnv, cn[j]), (uint32_t)j));
        }
        if (exact_rows && lane == 0) atomicAdd(exact_rows, 1u);
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) best = umin64(best, shfl_xor_u64(best, sft));
    if (lane == 0) {
        const float dis = ordered_to_f32((uint32_t)(best >> 32));
        const bool miss = !(dis < FLT_MAX_F);      // the reference's heap starts at FLT_MAX and admits only dis < top (Heap.h:76-78)
        cdis[q] = miss ? FLT_MAX_F : dis;
        keys[q] = miss ? -1 : (int64_t)(uint32_t)best;
    }
}

}  // namespace

bool coarse_screen_shape_ok(int nlist, int d, int nprobe) {
    return d >= 4 && d <= 128 && d % 4 == 0 && nlist % 64 == 0 && nlist >= 256 && nlist <= (1 << 20) && nprobe >= 2 && nprobe <= 128 &&
           nprobe * 2 <= nlist;      // (the cut pool needs nprobe distinct columns)
}

void launch_screen_prep(const float* x, const float* mu, int64_t n, int d, float scale, void* out_half, float* norms, float* norms_c,
                        unsigned char* flags, hipStream_t s) {
    if (n <= 0) return;
    const int64_t n_pad = (n + 127) / 128 * 128;
    hipLaunchKernelGGL(screen_prep_kernel, dim3((unsigned)(n_pad / kPrepRows)), dim3(128), (size_t)kPrepRows * (d + 4) * sizeof(float), s, x, mu, n, d,
                       (d + 15) / 16, scale, reinterpret_cast<_Float16*>(out_half), norms, norms_c, flags);
}

// keep_ws: the kept-column lists of the matrix path, or -- matrix-free -- the rows' bitmaps, bounds and flags
size_t coarse_screen_keep_bytes(int64_t nq, int nlist) {
    const size_t lists = (size_t)nq * (kKeepCap * sizeof(uint32_t) + sizeof(uint16_t));
    const size_t free_ = (size_t)nq * ((size_t)(nlist >> 5) * 4 + 4 + 1) + 64;
    return std::max(lists, free_);
}
// the matrix-free form: a pool of 32 class minima per (super-)range of 512 x nsub columns, at most 1024 per row, at least
// 2 nprobe of them; round 5: up to 16 384 lists and 64 probes; round 6: up to 2^20 lists (nsub ranges per workgroup) and 128 probes
// (VLQ_COARSE_MATRIX_FREE_WIDE=0: round 5's limits, the half-matrix form beyond them)
static int matrix_free_nsub(int nlist) {
    const int nranges = ((nlist >> 5) + 15) / 16;
    return (nranges + 31) / 32;
}
bool coarse_screen_matrix_free_ok(int nlist, int nprobe) {
    static const bool off = getenv("VLQ_COARSE_MATRIX") != nullptr;      // A/B: the half matrix of rounds 3-4
    static const bool wide = !(getenv("VLQ_COARSE_MATRIX_FREE_WIDE") && atoi(getenv("VLQ_COARSE_MATRIX_FREE_WIDE")) == 0);
    if (off || nlist < 1024 || (nlist & 63)) return false;
    if (!wide && (nlist > 16384 || nprobe > 64)) return false;
    const int nranges = ((nlist >> 5) + 15) / 16;
    const int nsuper = (nranges + matrix_free_nsub(nlist) - 1) / matrix_free_nsub(nlist);
    return nlist <= (1 << 20) && nprobe <= 128 && nprobe * 2 <= 32 * nsuper;
}

void launch_coarse_screened(const float* q, const void* q_half, const unsigned char* q_flags, const float* c, const void* c_half,
                            const float* qn, const float* cn, const float* qn_c, const float* cn_c,
                            float* approx /* [roundup128(nq)][nlist] halves */, float* tmin_ws /* nlist > 8192: [nq][nlist / 64] */,
                            void* keep_ws, int64_t nq,
                            int nlist, int d, int nprobe, float scale, float cmax, float cmax0, float* cdis, int64_t* keys,
                            unsigned long long* kept_total, unsigned int* exact_rows, hipStream_t s, OrderHist oh, bool* hist_done) {
    if (hist_done) *hist_done = false;
    if (nq <= 0) return;
    const int ks = (d + 15) / 16;
    const float inv_s2 = 1.f / (scale * scale);                // a power of two
    // the matrix holds half(sd * approximate distance), sd the power of two that puts 4 C^2 -- every distance of a query no
    // longer than the longest centroid -- just inside the half range
    int e = 0;
    (void)frexpf(65504.f / (4.04f * cmax * cmax), &e);
    const float sd = ldexpf(1.f, std::max(-120, std::min(120, e - 1)));
    _Float16* ah = reinterpret_cast<_Float16*>(approx);
    const _Float16* qh = reinterpret_cast<const _Float16*>(q_half);
    const _Float16* ch = reinterpret_cast<const _Float16*>(c_half);
    dim3 grid((unsigned)((nq + 127) / 128), (unsigned)((nlist + 127) / 128));
    if (tmin_ws && coarse_screen_matrix_free_ok(nlist, nprobe)) {
        // Matrix-free (round 5): (1) the approximate distances' 64-column tile minima, fp32, nothing else stored; (2) per row
        // the bound from the nprobe-th smallest tile minimum; (3) the approximate distances once more -- 4 us of f16 MFMA --
        // leaving one bit per element under the bound; (4) the exact stage on the marked columns.  Rounds 3-4 wrote the
        // matrix as halves and read it back: 82 MB each way at 10 000 x 4096, which is what both kernels were bound by.
        const float c_sub = 1.220703125e-04f /* 2^-13 */ * sqrtf((float)d) / scale * 1.001f;
        char* wsb = reinterpret_cast<char*>(keep_ws);
        uint32_t* bits = reinterpret_cast<uint32_t*>(wsb);
        float* trow = reinterpret_cast<float*>(wsb + (((size_t)nq * (nlist >> 5) * 4 + 15) & ~(size_t)15));
        unsigned char* rflag = reinterpret_cast<unsigned char*>(trow + nq);
        // a range = 16 blocks of 32 centroids (512 columns); the bound's pool = 32 classes per range
        const int nb_range = 16;
        const int nranges = ((nlist >> 5) + nb_range - 1) / nb_range;
        const int nsub = matrix_free_nsub(nlist);                   // ranges per workgroup (1 up to 16 384 lists)
        const int nsuper = (nranges + nsub - 1) / nsub;
        const int npool = 32 * nsuper;                             // <= 1024
        float* pool = tmin_ws;                                     // [nq][npool]  (the caller sized it for nlist / 16 + 32 floats per row)
        dim3 sgridq((unsigned)((nq + 255) / 256), (unsigned)nsuper);       // a workgroup = 4 waves x 64 queries
        const float m2 = -2.f * inv_s2;
#define VLQ_STR0(K) hipLaunchKernelGGL((coarse_f16_stream_kernel<K, 0>), sgridq, dim3(256), 0, s, qh, ch, cn_c, nq, nlist, nb_range, m2, pool, npool, \
                                       (const float*)nullptr, (uint32_t*)nullptr, nsub)
#define VLQ_STR1(K) hipLaunchKernelGGL((coarse_f16_stream_kernel<K, 1>), sgridq, dim3(256), 0, s, qh, ch, cn_c, nq, nlist, nb_range, m2, (float*)nullptr, npool, \
                                       (const float*)trow, bits, nsub)
#define VLQ_KS(M) switch (ks) { case 1: M(1); break; case 2: M(2); break; case 3: M(3); break; case 4: M(4); break; case 5: M(5); break; \
                                case 6: M(6); break; case 7: M(7); break; default: M(8); break; }
        VLQ_KS(VLQ_STR0)
        dim3 sgrid((unsigned)((nq + 3) / 4)), block(256);
#define VLQ_CUT(MT) hipLaunchKernelGGL(coarse_screen_cut_kernel<MT>, sgrid, block, 0, s, pool, nq, npool, nprobe, qn_c, qn, q_flags, cmax, cmax0, \
                                       c_sub, trow, rflag)
        if (npool <= 64) VLQ_CUT(1); else if (npool <= 128) VLQ_CUT(2); else if (npool <= 256) VLQ_CUT(4); else if (npool <= 512) VLQ_CUT(8); else VLQ_CUT(16);
        VLQ_KS(VLQ_STR1)
        if (nprobe <= 64)
            hipLaunchKernelGGL((coarse_screen_exact_kernel<1, true>), sgrid, block, 0, s, (const uint32_t*)nullptr, (const uint16_t*)nullptr, nq, nlist,
                               nprobe, cdis, keys, q, c, qn, cn, d, kept_total, (const uint32_t*)bits, (const unsigned char*)rflag, exact_rows, oh);
        else
            hipLaunchKernelGGL((coarse_screen_exact_kernel<2, true>), sgrid, block, 0, s, (const uint32_t*)nullptr, (const uint16_t*)nullptr, nq, nlist,
                               nprobe, cdis, keys, q, c, qn, cn, d, kept_total, (const uint32_t*)bits, (const unsigned char*)rflag, exact_rows, oh);
        if (hist_done) *hist_done = oh.hist != nullptr;
#undef VLQ_STR0
#undef VLQ_STR1
#undef VLQ_KS
#undef VLQ_CUT
        return;
    }
    float* tmin = nlist > 8192 ? tmin_ws : nullptr;
#define VLQ_F16G(K) hipLaunchKernelGGL(coarse_f16_dist_kernel<K>, grid, dim3(256), 0, s, qh, ch, qn_c, cn_c, ah, nq, nlist, inv_s2, sd, tmin)
    switch (ks) {
        case 1: VLQ_F16G(1); break;
        case 2: VLQ_F16G(2); break;
        case 3: VLQ_F16G(3); break;
        case 4: VLQ_F16G(4); break;
        case 5: VLQ_F16G(5); break;
        case 6: VLQ_F16G(6); break;
        case 7: VLQ_F16G(7); break;
        default: VLQ_F16G(8); break;
    }
#undef VLQ_F16G
    // (2^-13: as if the MFMA flushed subnormal half operands to zero -- 2^-14 per component and side --, which covers rounding
    // them, 2^-25, with room to spare; either way this term is noise next to the first one on any data worth screening)
    const float c_sub = 1.220703125e-04f /* 2^-13 */ * sqrtf((float)d) / scale * 1.001f;
    uint32_t* keep = reinterpret_cast<uint32_t*>(keep_ws);
    uint16_t* nkeep = reinterpret_cast<uint16_t*>(keep + (size_t)nq * kKeepCap);
    dim3 sgrid((unsigned)((nq + 3) / 4)), block(256);
#define VLQ_SCR(NV, MT) hipLaunchKernelGGL((coarse_screen_keep_kernel<NV, MT>), sgrid, block, 0, s, ah, nq, nlist, nprobe, qn_c, qn, q_flags, \
                                           cmax, cmax0, c_sub, 1.f / sd, keep, nkeep, exact_rows)
#define VLQ_SCRT(MT) hipLaunchKernelGGL(coarse_screen_keep_tiled_kernel<MT>, sgrid, block, 0, s, ah, tmin, nq, nlist, nprobe, qn_c, qn, q_flags, \
                                        cmax, cmax0, c_sub, 1.f / sd, keep, nkeep, exact_rows)
    // cut pool: the 2 (nprobe <= 64) or 4 (<= 128) smallest elements / tile minima of every lane
    if (nprobe <= 64) {
        if (nlist <= 1024) VLQ_SCR(4, 2);
        else if (nlist <= 2048) VLQ_SCR(8, 2);
        else if (nlist <= 4096) VLQ_SCR(16, 2);
        else if (nlist <= 8192) VLQ_SCR(32, 2);
        else VLQ_SCRT(2);
        hipLaunchKernelGGL(coarse_screen_exact_kernel<1>, sgrid, block, 0, s, keep, nkeep, nq, nlist, nprobe, cdis, keys, q, c, qn, cn, d,
                           kept_total);
    } else {
        if (nlist <= 1024) VLQ_SCR(4, 4);
        else if (nlist <= 2048) VLQ_SCR(8, 4);
        else if (nlist <= 4096) VLQ_SCR(16, 4);
        else if (nlist <= 8192) VLQ_SCR(32, 4);
        else VLQ_SCRT(4);
        hipLaunchKernelGGL(coarse_screen_exact_kernel<2>, sgrid, block, 0, s, keep, nkeep, nq, nlist, nprobe, cdis, keys, q, c, qn, cn, d,
                           kept_total);
    }
#undef VLQ_SCR
#undef VLQ_SCRT
}

bool coarse_screen_nn_shape_ok(int nlist, int d) {
    return d >= 4 && d <= 128 && d % 4 == 0 && nlist % 64 == 0 && nlist >= 256 && nlist <= (1 << 20);
}

void launch_coarse_screened_nn(const float* q, const void* q_half, const unsigned char* q_flags, const float* c, const void* c_half,
                               const float* qn, const float* cn, const float* qn_c, const float* cn_c, float* tmin_ws, int64_t nq, int nlist,
                               int d, float scale, float cmax, float cmax0, float* cdis, int64_t* keys, unsigned int* exact_rows,
                               hipStream_t s) {
    if (nq <= 0) return;
    const int ks = (d + 15) / 16;
    const float inv_s2 = 1.f / (scale * scale);
    const _Float16* qh = reinterpret_cast<const _Float16*>(q_half);
    const _Float16* ch = reinterpret_cast<const _Float16*>(c_half);
    dim3 grid((unsigned)((nq + 127) / 128), (unsigned)((nlist + 127) / 128));
#define VLQ_F16M(K) hipLaunchKernelGGL(coarse_f16_dist_kernel<K>, grid, dim3(256), 0, s, qh, ch, qn_c, cn_c, (_Float16*)nullptr, nq, nlist, inv_s2, \
                                       1.f, tmin_ws)
    switch (ks) {
        case 1: VLQ_F16M(1); break;
        case 2: VLQ_F16M(2); break;
        case 3: VLQ_F16M(3); break;
        case 4: VLQ_F16M(4); break;
        case 5: VLQ_F16M(5); break;
        case 6: VLQ_F16M(6); break;
        case 7: VLQ_F16M(7); break;
        default: VLQ_F16M(8); break;
    }
#undef VLQ_F16M
    const float c_sub = 1.220703125e-04f /* 2^-13 */ * sqrtf((float)d) / scale * 1.001f;
    hipLaunchKernelGGL(coarse_screen_nn_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, tmin_ws, nq, nlist, cdis, keys, q, c, qn, cn,
                       qn_c, q_flags, d, cmax, cmax0, c_sub, exact_rows);
}

void preload_coarse_screen_kernels() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(screen_prep_kernel));
}

}  // namespace vlq
