// HIP kernels of the VLQ (vector and line quantization) path for gfx950.
//
// Stage map (reference CUDA kernel -> kernel here).  The reference leaves the operation
// order to nvcc's contraction and ties to unstable sorts; here every fp32 operation is an
// explicit unfused multiply/add/divide in the order written, ties go to the lowest index
// (VLQ section):
//   get1BinKernel_nms        gpu/GpuIndexFlat.cu:433-557         -> line_assign_kernel
//   assignLambdaKernel       gpu/GpuIndexFlat.cu:559-602         -> lambda_quantize_kernel
//   calResidual              gpu/GpuIndexFlat.cu:1092-1129        -> line_residual_kernel
//   sumAlongRowsWithOrder2   gpu/impl/BroadcastSum.cu:477-560     -> line_select_kernel
//   pqScanPrecomputedMultiPassGraph + pass1/pass2 select
//                            gpu/impl/PQScanMultiPassPrecomputed.cu:675-811,
//                            IVFUtilsSelect1.cu, IVFUtilsSelect2.cu:398-569 -> line_scan_kernel
#include "line.h"
#include "scan_common.cuh"
#include "sse_order.cuh"
#include "wave_topk.cuh"

namespace vlq {

#define FLT_MAX_F 3.402823466e+38f

// ---------------------------------------------------------------------------
// line assignment: one wave per vector, lanes over the edges of its nearest centroid
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void line_assign_kernel(
    const float* __restrict__ x, int64_t n, int d, const float* __restrict__ coarse,
    const int64_t* __restrict__ nearest, const int32_t* __restrict__ edge_info,
    const float* __restrict__ edge_dist, int nedge, int32_t* __restrict__ line_id,
    float* __restrict__ lambdaf) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][d]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + wave;
    if (v >= n) return;
    const int64_t A = nearest[v];
    if (A < 0) { if (lane == 0) { line_id[v] = -1; lambdaf[v] = 0.f; } return; }
    float* xs = sm + wave * d;
    for (int c = lane; c < d; c += 64) xs[c] = x[v * d + c];
    __builtin_amdgcn_wave_barrier();
    const float* ca = coarse + A * d;
    const float b2 = l2sqr_sse_order([&](int c) { return xs[c]; }, [&](int c) { return ca[c]; }, d);
    // per lane: best edge overall and best edge with 0 <= lambda <= 1, ties to the lowest edge
    int be = -1, bi = -1;
    float bd = 0.f, bl = 0.f, bdi = 0.f, bli = 0.f;
    for (int e = lane; e < nedge; e += 64) {
        const int s = edge_info[A * nedge + e];
        const float* cs = coarse + (int64_t)s * d;
        const float a2 = l2sqr_sse_order([&](int c) { return xs[c]; }, [&](int c) { return cs[c]; }, d);
        const float c2 = edge_dist[A * nedge + e];
        const float amb = __fsub_rn(__fsub_rn(a2, b2), c2);                 // a2 - b2 - c2
        const float l = __fdiv_rn(__fmul_rn(-0.5f, amb), c2);               // project(), triangle.cuh:86-87
        const float d2 = __fadd_rn(__fadd_rn(b2, __fmul_rn(__fmul_rn(l, l), c2)), __fmul_rn(l, amb));   // dist2()
        if (be < 0 || d2 < bd) { be = e; bd = d2; bl = l; }
        if (l >= 0.f && l <= 1.f && (bi < 0 || d2 < bdi)) { bi = e; bdi = d2; bli = l; }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
        int oe = __shfl_xor(be, sft, 64);
        float od = __shfl_xor(bd, sft, 64), ol = __shfl_xor(bl, sft, 64);
        if (oe >= 0 && (be < 0 || od < bd || (od == bd && oe < be))) { be = oe; bd = od; bl = ol; }
        oe = __shfl_xor(bi, sft, 64);
        od = __shfl_xor(bdi, sft, 64);
        ol = __shfl_xor(bli, sft, 64);
        if (oe >= 0 && (bi < 0 || od < bdi || (od == bdi && oe < bi))) { bi = oe; bdi = od; bli = ol; }
    }
    if (lane == 0) {
        const int e = bi >= 0 ? bi : be;
        line_id[v] = (int32_t)(A * nedge + e);
        lambdaf[v] = bi >= 0 ? bli : bl;
    }
}

void launch_line_assign(const float* x, int64_t n, int d, const float* coarse, const int64_t* nearest,
                        const int32_t* edge_info, const float* edge_dist, int nedge, int32_t* line_id,
                        float* lambdaf, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(line_assign_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256),
                       (size_t)4 * d * sizeof(float), s, x, n, d, coarse, nearest, edge_info, edge_dist,
                       nedge, line_id, lambdaf);
}

// nearest scalar of the codebook, first minimum
__global__ void lambda_quantize_kernel(const float* __restrict__ lambdaf, int64_t n,
                                       const float* __restrict__ lambda_info, int nlambda,
                                       uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = lambdaf[i];
    int bi = 0;
    float bd = FLT_MAX_F;
    for (int j = 0; j < nlambda; j++) {
        const float t = __fsub_rn(v, lambda_info[j]);
        const float dd = __fmul_rn(t, t);
        if (dd < bd) { bd = dd; bi = j; }
    }
    out[i] = (uint8_t)bi;
}

void launch_lambda_quantize(const float* lambdaf, int64_t n, const float* lambda_info, int nlambda,
                            uint8_t* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(lambda_quantize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       lambdaf, n, lambda_info, nlambda, out);
}

// x - ((1-l) c_A + l c_s)
__global__ void line_residual_kernel(const float* __restrict__ x, int64_t n, int d,
                                     const float* __restrict__ coarse,
                                     const int32_t* __restrict__ edge_info, int nedge,
                                     const int32_t* __restrict__ line_id,
                                     const uint8_t* __restrict__ lambda,
                                     const float* __restrict__ lambda_info, float* __restrict__ res) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * d) return;
    const int64_t i = e / d;
    const int j = (int)(e % d);
    const int32_t line = line_id[i];
    if (line < 0) { res[e] = 0.f; return; }
    const int A = line / nedge;
    const int s = edge_info[line];
    const float la = lambda_info[lambda[i]];
    const float oml = __fsub_rn(1.f, la);
    const float anchor = __fadd_rn(__fmul_rn(oml, coarse[(int64_t)A * d + j]), __fmul_rn(la, coarse[(int64_t)s * d + j]));
    res[e] = __fsub_rn(x[e], anchor);
}

void launch_line_residuals(const float* x, int64_t n, int d, const float* coarse, const int32_t* edge_info,
                           int nedge, const int32_t* line_id, const uint8_t* lambda,
                           const float* lambda_info, float* res, hipStream_t s) {
    if (n <= 0) return;
    const int64_t tot = n * d;
    hipLaunchKernelGGL(line_residual_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, x, n,
                       d, coarse, edge_info, nedge, line_id, lambda, lambda_info, res);
}

// ---------------------------------------------------------------------------
// line select: one wave per query over the nprobe x nedge candidate lines
// ---------------------------------------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void line_select_kernel(
    const float* __restrict__ dist, int64_t nq, int nlist, const int64_t* __restrict__ keys,
    int nprobe, const int32_t* __restrict__ edge_info, const float* __restrict__ edge_dist, int nedge,
    int w1, int32_t* __restrict__ sel_line, float* __restrict__ sel_b2, float* __restrict__ sel_g,
    const int64_t* __restrict__ line_off, const int64_t* __restrict__ line_len, int max_line_codes,
    LineMeta* __restrict__ sel_meta, int32_t* __restrict__ sel_cnt) {
    __shared__ u64 queue[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;
    WaveSelect<KPL> sel;
    sel.init(w1, queue[wave], lane);
    const float* row = dist + q * nlist;
    const int64_t* kq = keys + q * nprobe;
    const int num = nprobe * nedge;
    for (int i0 = 0; i0 < num; i0 += 64) {
        const int i = i0 + lane;
        bool valid = i < num;
        float key = 0.f;
        if (valid) {
            const int64_t c = kq[i / nedge];
            valid = c >= 0;
            if (valid) {
                const int e = i % nedge;
                const int s = edge_info[c * nedge + e];
                const float a2 = row[s], b2 = row[c], c2 = edge_dist[c * nedge + e];
                const float g = __fsub_rn(a2, b2);
                const float t = __fsub_rn(g, c2);
                // BroadcastSum.cu:503-507: beyond the near end -> distance to c, else to the line
                key = (t > 0.f) ? b2 : __fsub_rn(b2, __fdiv_rn(__fmul_rn(__fmul_rn(0.25f, t), t), c2));
            }
        }
        sel.offer(key, (uint32_t)i, valid);
    }
    sel.flush();
    // (1) the w1 winners in ascending key order -- the order sumAlongRowsWithOrder2 emits them in
    // (BroadcastSum.cu:538-553) and therefore the order of the scan's output array: scan positions,
    // the tie rule of the final top-k, follow THIS order
    uint32_t len_r[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int w = r * 64 + lane;
        const u64 k64 = sel.best[r];
        int32_t line = -1;
        float b2 = 0.f, g = 0.f;
        uint32_t len = 0;
        if (w < w1 && k64 != kMaxKey) {
            const int i = (int)(uint32_t)k64;
            const int64_t c = kq[i / nedge];
            const int e = i % nedge;
            const int s = edge_info[c * nedge + e];
            line = (int32_t)(c * nedge + e);
            b2 = row[c];
            g = __fsub_rn(row[s], b2);
            if (sel_meta) {
                int64_t l64 = line_len ? line_len[line] : line_off[line + 1] - line_off[line];
                if (l64 > max_line_codes) l64 = max_line_codes;
                len = (uint32_t)l64;
            }
        }
        if (w < w1) {
            sel_line[q * w1 + w] = line;
            sel_b2[q * w1 + w] = b2;
            sel_g[q * w1 + w] = g;
        }
        len_r[r] = len;
    }
    if (!sel_meta) return;
    // (2) compact records for the 16-byte scan: the non-empty lines, each with its scan position and
    // its rank in the emitted order, but stored in candidate order (probe rank, edge) so that lines
    // sharing an anchor centroid are neighbours and the scan builds the anchor's table once per group.
    // Sort key = candidate index << 40 | rank << 30 | position (w1, codes per line <= 1024; the host
    // checks nprobe * nedge < 2^24).
    WaveSelect<KPL> ord;
    ord.init(w1, queue[wave], lane);
    uint32_t run_pos = 0, run_cnt = 0;
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const bool keep = len_r[r] > 0;
        const u64 mask = __ballot(keep);
        const uint32_t rank = run_cnt + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        uint32_t inc = len_r[r];
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t o = __shfl_up(inc, sft, 64);
            if (lane >= sft) inc += o;
        }
        const uint32_t pos0 = run_pos + inc - len_r[r];
        run_pos += __shfl(inc, 63, 64);
        run_cnt += (uint32_t)__popcll(mask);
        const u64 i = (uint32_t)sel.best[r];
        ord.offer_key((i << 40) | ((u64)rank << 30) | pos0, keep);
    }
    ord.flush();
    // only non-empty lines were offered: sorted elements 0 .. run_cnt-1 are the records
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int w = r * 64 + lane;
        const u64 k64 = ord.best[r];
        if (w >= (int)run_cnt || k64 == kMaxKey) continue;
        const int i = (int)(k64 >> 40);
        const int64_t c = kq[i / nedge];
        const int e = i % nedge;
        const int64_t line = c * nedge + e;
        LineMeta m;
        m.off = line_off[line];
        int64_t len = line_len ? line_len[line] : line_off[line + 1] - m.off;
        if (len > max_line_codes) len = max_line_codes;
        m.len = (int32_t)len;
        m.line = (int32_t)line;
        m.s = edge_info[line];
        m.c2 = edge_dist[line];
        m.b2 = row[c];
        m.g = __fsub_rn(row[m.s], m.b2);
        m.pos0 = (uint32_t)k64 & ((1u << 30) - 1u);
        m.rank = (int32_t)((k64 >> 30) & 1023u);
        m.anchor = (int32_t)c;
        m.pad1 = 0;
        sel_meta[q * w1 + w] = m;
    }
    if (lane == 0) sel_cnt[q] = (int32_t)run_cnt;
}

void launch_line_select(const float* dist, int64_t nq, int nlist, const int64_t* keys, int nprobe,
                        const int32_t* edge_info, const float* edge_dist, int nedge, int w1,
                        int32_t* sel_line, float* sel_b2, float* sel_g, hipStream_t s,
                        const int64_t* line_off, const int64_t* line_len, int max_line_codes,
                        LineMeta* sel_meta, int32_t* sel_cnt) {
    if (nq <= 0) return;
    dim3 grid((unsigned)((nq + 3) / 4)), block(256);
#define VLQ_LS(K) hipLaunchKernelGGL(line_select_kernel<K>, grid, block, 0, s, dist, nq, nlist, keys, nprobe, \
                                     edge_info, edge_dist, nedge, w1, sel_line, sel_b2, sel_g, line_off,      \
                                     line_len, max_line_codes, sel_meta, sel_cnt)
    if (w1 <= 64) VLQ_LS(1);
    else if (w1 <= 256) VLQ_LS(4);
    else VLQ_LS(16);
#undef VLQ_LS
}

// Two LUTs, one address: T23 at LDS byte 0, T4 at byte 16384 (the kernel's dynamic LDS starts at
// LDS address 0: no static LDS).  One SDWA op extracts a code byte and scales it by 4; the T4
// read takes its own register, the T23 read lands in the address register.
#define VLQ_L16_LO(W0, W1)                                                                      \
    asm volatile(                                                                          \
        "v_lshlrev_b32_sdwa %0, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %1, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %2, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %3, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "v_lshlrev_b32_sdwa %4, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %5, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %6, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %7, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b32 %8, %0 offset:16384\n\t" \
        "ds_read_b32 %0, %0 offset:0\n\t" \
        "ds_read_b32 %9, %1 offset:17408\n\t" \
        "ds_read_b32 %1, %1 offset:1024\n\t" \
        "ds_read_b32 %10, %2 offset:18432\n\t" \
        "ds_read_b32 %2, %2 offset:2048\n\t" \
        "ds_read_b32 %11, %3 offset:19456\n\t" \
        "ds_read_b32 %3, %3 offset:3072\n\t" \
        "ds_read_b32 %12, %4 offset:20480\n\t" \
        "ds_read_b32 %4, %4 offset:4096\n\t" \
        "ds_read_b32 %13, %5 offset:21504\n\t" \
        "ds_read_b32 %5, %5 offset:5120\n\t" \
        "ds_read_b32 %14, %6 offset:22528\n\t" \
        "ds_read_b32 %6, %6 offset:6144\n\t" \
        "ds_read_b32 %15, %7 offset:23552\n\t" \
        "ds_read_b32 %7, %7 offset:7168\n\t" \
        "s_waitcnt lgkmcnt(0)"                                                                                 \
        : "=&v"(va[0]), "=&v"(va[1]), "=&v"(va[2]), "=&v"(va[3]), "=&v"(va[4]), "=&v"(va[5]), "=&v"(va[6]), "=&v"(va[7]), "=&v"(vb[0]), "=&v"(vb[1]), "=&v"(vb[2]), "=&v"(vb[3]), "=&v"(vb[4]), "=&v"(vb[5]), "=&v"(vb[6]), "=&v"(vb[7])                                                                               \
        : "v"(W0), "v"(W1), "v"(two)                                                       \
        : "memory")
#define VLQ_L16_HI(W0, W1)                                                                      \
    asm volatile(                                                                          \
        "v_lshlrev_b32_sdwa %0, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %1, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %2, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %3, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "v_lshlrev_b32_sdwa %4, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %5, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %6, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %7, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b32 %8, %0 offset:24576\n\t" \
        "ds_read_b32 %0, %0 offset:8192\n\t" \
        "ds_read_b32 %9, %1 offset:25600\n\t" \
        "ds_read_b32 %1, %1 offset:9216\n\t" \
        "ds_read_b32 %10, %2 offset:26624\n\t" \
        "ds_read_b32 %2, %2 offset:10240\n\t" \
        "ds_read_b32 %11, %3 offset:27648\n\t" \
        "ds_read_b32 %3, %3 offset:11264\n\t" \
        "ds_read_b32 %12, %4 offset:28672\n\t" \
        "ds_read_b32 %4, %4 offset:12288\n\t" \
        "ds_read_b32 %13, %5 offset:29696\n\t" \
        "ds_read_b32 %5, %5 offset:13312\n\t" \
        "ds_read_b32 %14, %6 offset:30720\n\t" \
        "ds_read_b32 %6, %6 offset:14336\n\t" \
        "ds_read_b32 %15, %7 offset:31744\n\t" \
        "ds_read_b32 %7, %7 offset:15360\n\t" \
        "s_waitcnt lgkmcnt(0)"                                                                                 \
        : "=&v"(va[0]), "=&v"(va[1]), "=&v"(va[2]), "=&v"(va[3]), "=&v"(va[4]), "=&v"(va[5]), "=&v"(va[6]), "=&v"(va[7]), "=&v"(vb[0]), "=&v"(vb[1]), "=&v"(vb[2]), "=&v"(vb[3]), "=&v"(vb[4]), "=&v"(vb[5]), "=&v"(vb[6]), "=&v"(vb[7])                                                                               \
        : "v"(W0), "v"(W1), "v"(two)                                                       \
        : "memory")

// line scan specialised for 16-byte codes (M = 16, ksub = 256): same arithmetic as
// line_scan_kernel below, organised like the IVFPQ scan16 kernel --
//   * the selected lines arrive grouped by anchor centroid and without empty lines
//     (LineMeta, written by line_select_kernel): T23 = term2[c] - 2<q, .> is rebuilt only when
//     the anchor changes, the anchor's term2 row stays in registers, and a line costs one
//     16 KB row read (term2[s]) and one 16 KB LDS table (T4 = term2[s] - term2[c]) instead of two;
//   * the next line's row, its first code chunk and lambda bytes are requested before the
//     current line is scanned; per-query -2<q, cent> (16 entries per thread) lives in registers;
//   * one SDWA op per code byte serves both table lookups.
template <int KPL>
__global__ __launch_bounds__(256, KPL <= 4 ? 4 : 2) void line16_scan_kernel(LineScanArgs a, int queue_off) {
    constexpr int E = 4096, NT = 256, NI = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* t23 = reinterpret_cast<float*>(smraw);                    // [E] at LDS byte 0
    float* t4 = t23 + E;                                             // [E] at LDS byte 16384
    float* lamtab = t4 + E;                                          // [256]
    u64* queue = reinterpret_cast<u64*>(smraw + queue_off);          // [4][64]
    uint32_t* cum = reinterpret_cast<uint32_t*>(queue + 4 * 64);     // [w1+1] scan position of the rank-th line
    uint16_t* wmap = reinterpret_cast<uint16_t*>(cum + a.w1 + 1);    // [w1] rank -> record index

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
    const int64_t q = blockIdx.x;
    const int cnt = a.sel_cnt[q];
    const uint4* mq = reinterpret_cast<const uint4*>(a.sel_meta + q * a.w1);   // 3 x 16 bytes per record

    float4 m2q[NI];                       // -2 <q_m, cent_mj>, entries 4*(i*256+t) .. +3
    {
        const float4* qt = reinterpret_cast<const float4*>(a.qtab + q * E);
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const float4 v = qt[i * NT + t];
            m2q[i] = make_float4(__fmul_rn(-2.f, v.x), __fmul_rn(-2.f, v.y), __fmul_rn(-2.f, v.z), __fmul_rn(-2.f, v.w));
        }
    }
    lamtab[t] = a.lambda_info[t];         // padded to 256 entries by the host
    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);

    float4 t2c[NI], ts[NI];
    uint4 c0 = make_uint4(0, 0, 0, 0);
    uint32_t l0 = 0;
    auto prefetch = [&](const uint4 m0, const uint4 m1) __attribute__((always_inline)) {
        const int64_t off = (int64_t)(((uint64_t)m0.y << 32) | m0.x);
        const uint32_t len = m0.z;
        const int32_t s = (int32_t)m1.x;
        const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)s * E);
#pragma unroll
        for (int i = 0; i < NI; i++) ts[i] = src[i * NT + t];
        const uint32_t j = min((uint32_t)t, len - 1);
        c0 = reinterpret_cast<const uint4*>(a.codes)[off + j];
        l0 = a.lambdas[off + j];
    };
    uint4 ma0 = make_uint4(0, 0, 0, 0), ma1 = ma0, ma2 = ma0, mb0 = ma0, mb1 = ma0, mb2 = ma0;
    if (cnt > 0) {
        ma0 = mq[0]; ma1 = mq[1]; ma2 = mq[2];
        const int w1c = min(1, cnt - 1);
        mb0 = mq[3 * w1c]; mb1 = mq[3 * w1c + 1]; mb2 = mq[3 * w1c + 2];
        prefetch(ma0, ma1);
    }
    int cprev = -1;
    uint32_t total = 0;
    for (int w = 0; w < cnt; w++) {
        const int64_t off = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane(ma0.y) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane(ma0.x));
        const uint32_t len = __builtin_amdgcn_readfirstlane(ma0.z);
        const int line = __builtin_amdgcn_readfirstlane(ma0.w);
        const float c2 = __uint_as_float(__builtin_amdgcn_readfirstlane(ma1.y));
        const float b2 = __uint_as_float(__builtin_amdgcn_readfirstlane(ma1.z));
        const float g = __uint_as_float(__builtin_amdgcn_readfirstlane(ma1.w));
        // scan position of the line's first code and the line's rank, both in the emitted (ascending
        // key) order of the line select; the records themselves come grouped by anchor
        const uint32_t pos0 = __builtin_amdgcn_readfirstlane(ma2.x);
        const int rank = __builtin_amdgcn_readfirstlane(ma2.y);
        const int c = line / a.nedge;
        if (t == 0) { cum[rank] = pos0; wmap[rank] = (uint16_t)w; }
        __syncthreads();                         // previous line fully scanned
        __builtin_amdgcn_s_setprio(2);           // table build + next line's loads first (see scan16.hip)
        if (c != cprev) {                        // new anchor: its row into registers, T23 into LDS
            const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)c * E);
#pragma unroll
            for (int i = 0; i < NI; i++) t2c[i] = src[i * NT + t];
#pragma unroll
            for (int i = 0; i < NI; i++) {
                float4 v;
                v.x = __fadd_rn(t2c[i].x, m2q[i].x); v.y = __fadd_rn(t2c[i].y, m2q[i].y);
                v.z = __fadd_rn(t2c[i].z, m2q[i].z); v.w = __fadd_rn(t2c[i].w, m2q[i].w);
                reinterpret_cast<float4*>(t23)[i * NT + t] = v;
            }
            cprev = c;
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
            float4 v;
            v.x = __fsub_rn(ts[i].x, t2c[i].x); v.y = __fsub_rn(ts[i].y, t2c[i].y);
            v.z = __fsub_rn(ts[i].z, t2c[i].z); v.w = __fsub_rn(ts[i].w, t2c[i].w);
            reinterpret_cast<float4*>(t4)[i * NT + t] = v;
        }
        uint4 cc = c0;
        uint32_t lb = l0;
        ma0 = mb0; ma1 = mb1; ma2 = mb2;
        if (w + 1 < cnt) {
            prefetch(ma0, ma1);
            const int w2 = min(w + 2, cnt - 1);
            mb0 = mq[3 * w2]; mb1 = mq[3 * w2 + 1]; mb2 = mq[3 * w2 + 2];
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + off;
        const uint8_t* lp = a.lambdas + off;
        for (uint32_t j0 = (uint32_t)wave * 64; j0 < len; j0 += NT) {
            const uint32_t j = j0 + lane;
            const uint32_t jn = min(j + NT, len - 1);
            const uint4 cn = cp[jn];
            const uint32_t ln = lp[jn];
            const float l = lamtab[lb];
            // PQScanMultiPassPrecomputed.cu:783-811 as written: dist = term1 + la*term6 + (la*la-la)*term5,
            // then dist += term23[m] for m ascending; tmp += term4[m] from 0; result dist + la*tmp
            float dist = __fadd_rn(__fadd_rn(b2, __fmul_rn(l, g)), __fmul_rn(__fsub_rn(__fmul_rn(l, l), l), c2));
            float tmp = 0.f;
            {
                float va[8], vb[8];
                VLQ_L16_LO(cc.x, cc.y);
#pragma unroll
                for (int m = 0; m < 8; m++) { dist = __fadd_rn(dist, va[m]); tmp = __fadd_rn(tmp, vb[m]); }
            }
            {
                float va[8], vb[8];
                VLQ_L16_HI(cc.z, cc.w);
#pragma unroll
                for (int m = 0; m < 8; m++) { dist = __fadd_rn(dist, va[m]); tmp = __fadd_rn(tmp, vb[m]); }
            }
            dist = __fadd_rn(dist, __fmul_rn(l, tmp));
            // positions do not arrive in increasing order (records are grouped by anchor): the full
            // (distance, position) key decides among equal distances
            sel.template offer<false>(dist, pos0 + j, j < len);
            cc = cn;
            lb = ln;
        }
        total += len;
    }
    if (t == 0) cum[cnt] = total;

    ScanArgs em;                 // only the fields merge_and_emit reads
    em.k = a.k;
    em.nprobe = cnt > 0 ? cnt : 1;
    em.store_pairs = 0;
    em.ids = a.ids;
    em.D = a.D;
    em.I = a.I;
    if (cnt == 0 && t == 0) cum[1] = 0;
    merge_and_emit<KPL>(sel, smraw, cum, em, q, wave, lane, [&](int rank, int64_t& lkey, int64_t& loff) {
        const uint4 m0 = mq[3 * (int)wmap[rank]];
        lkey = (int64_t)(int32_t)m0.w;
        loff = (int64_t)(((uint64_t)m0.y << 32) | m0.x);
    });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)total);
}

// ---------------------------------------------------------------------------
// float16 look-up tables (GpuIndexIVFPQConfig::useFloat16LookupTables -- what the reference's VLQ
// drivers run with, gpu/test/deep1b16_query.cpp:239-243).  As in the reference: term 2 and term 3 are
// kept as half (impl/IVFPQ.cu:1442), the two tables are formed with HALF arithmetic
// (PQScanMultiPassPrecomputed.cu:54-75 add, :313-334 sub) and the looked-up entries are accumulated in
// float (:798-805).  A line then costs one 8 KB row instead of 16 KB -- the bytes that bound this scan.
// Same line order, same scan positions, same summation order as line16_scan_kernel.
// ---------------------------------------------------------------------------
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
union H8 { uint4 u; h16x2 h[4]; };

__global__ void to_half_kernel(const float* __restrict__ in, int64_t n, float scale, uint16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const _Float16 h = (_Float16)__fmul_rn(scale, in[i]);
    out[i] = __builtin_bit_cast(uint16_t, h);
}
void launch_to_half(const float* in, int64_t n, float scale, uint16_t* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, scale, out);
}

// One table of 4096 dwords at LDS byte 0: entry (m, j) = {T23h[m][j] (low half), T4h[m][j] (high half)}.  One SDWA
// op makes code byte * 4 and ONE ds_read_b32 per code byte serves both look-ups (two ds_read_u16 before: the LDS
// gather rate, not the 8 KB rows, was what held this kernel at 4.9 TB/s of fabric reads, profiles/r03_pmc_vlq.txt).
#define VLQ_HP_BLOCK(W0, W1, O)                                                                   \
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
        "ds_read_b32 %7, %7 offset:" #O "+7168\n\t" \
        "s_waitcnt lgkmcnt(0)"                                                                                 \
        : "=&v"(vp[0]), "=&v"(vp[1]), "=&v"(vp[2]), "=&v"(vp[3]), "=&v"(vp[4]), "=&v"(vp[5]), "=&v"(vp[6]), "=&v"(vp[7]) \
        : "v"(W0), "v"(W1), "v"(two)                                                       \
        : "memory")

template <int KPL>
__global__ __launch_bounds__(256, KPL <= 4 ? 5 : 2) void line16h_scan_kernel(LineScanArgs a, int queue_off) {
    constexpr int E = 4096, NT = 256, NI = 2;       // a row of 4096 halves = 512 x 16 bytes: two per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    uint4* tab = reinterpret_cast<uint4*>(smraw);                    // [E] dwords {T23h, T4h} at LDS byte 0
    float* lamtab = reinterpret_cast<float*>(smraw + 16384);         // [256]
    u64* queue = reinterpret_cast<u64*>(smraw + queue_off);          // [4][64]
    uint32_t* cum = reinterpret_cast<uint32_t*>(queue + 4 * 64);     // [w1+1] scan position of the rank-th line
    uint16_t* wmap = reinterpret_cast<uint16_t*>(cum + a.w1 + 1);    // [w1] rank -> record index

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (__builtin_amdgcn_groupstaticsize() != 0) return;             // the table offsets above are absolute
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
    const int64_t q = blockIdx.x;
    const int cnt = a.sel_cnt[q];
    const uint4* mq = reinterpret_cast<const uint4*>(a.sel_meta + q * a.w1);   // 3 x 16 bytes per record

    H8 q3[NI];                            // half(-2 <q_m, cent_mj>), entries 8*(i*256+t) .. +7
    {
        const uint4* qt = reinterpret_cast<const uint4*>(a.qtabh + q * E);
#pragma unroll
        for (int i = 0; i < NI; i++) q3[i].u = qt[i * NT + t];
    }
    lamtab[t] = a.lambda_info[t];         // padded to 256 entries by the host
    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);

    H8 t2c[NI], t23[NI], ts[NI];              // anchor row, T23h, far-end row of the next line
    uint4 c0 = make_uint4(0, 0, 0, 0);
    uint32_t l0 = 0;
    // a line record is 12 dwords: lane l < 12 holds dword l, fields are read with v_readlane (2 registers for two
    // records instead of 24).  A second row in flight per workgroup (rows requested two lines ahead) was measured
    // and is slower, as it is with fp32 tables: 5.54 against 5.24 ms at the C5 geometry.
    const uint32_t* mqw = reinterpret_cast<const uint32_t*>(mq);
    const int fl = lane < 12 ? lane : 0;
    auto rec_off = [&](uint32_t rec) __attribute__((always_inline)) {
        return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(rec, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane(rec, 0));
    };
    auto load_row = [&](uint32_t rec, H8 (&dst)[NI]) __attribute__((always_inline)) {
        const int32_t s = __builtin_amdgcn_readlane(rec, 4);
        const uint4* src = reinterpret_cast<const uint4*>(a.term2h + (size_t)s * E);
#pragma unroll
        for (int i = 0; i < NI; i++) dst[i].u = src[i * NT + t];
    };
    auto load_codes = [&](uint32_t rec) __attribute__((always_inline)) {
        const int64_t off = rec_off(rec);
        const uint32_t len = __builtin_amdgcn_readlane(rec, 2);
        const uint32_t j = min((uint32_t)t, len - 1);
        c0 = reinterpret_cast<const uint4*>(a.codes)[off + j];
        l0 = a.lambdas[off + j];
    };
    uint32_t mcur = 0, mnext = 0;
    if (cnt > 0) {
        mcur = mqw[fl];
        mnext = mqw[12 * min(1, cnt - 1) + fl];
        load_row(mcur, ts);
        load_codes(mcur);
    }
    int cprev = -1;
    uint32_t total = 0;
    for (int w = 0; w < cnt; w++) {
        const int64_t off = rec_off(mcur);
        const uint32_t len = __builtin_amdgcn_readlane(mcur, 2);
        const int line = __builtin_amdgcn_readlane(mcur, 3);
        const float c2 = __uint_as_float(__builtin_amdgcn_readlane(mcur, 5));
        const float b2 = __uint_as_float(__builtin_amdgcn_readlane(mcur, 6));
        const float g = __uint_as_float(__builtin_amdgcn_readlane(mcur, 7));
        const uint32_t pos0 = __builtin_amdgcn_readlane(mcur, 8);
        const int rank = __builtin_amdgcn_readlane(mcur, 9);
        const int c = line / a.nedge;
        if (t == 0) { cum[rank] = pos0; wmap[rank] = (uint16_t)w; }
        if (c != cprev) {                        // new anchor: its half row and T23h = hadd(term2h[c], term3h) in registers
            const uint4* src = reinterpret_cast<const uint4*>(a.term2h + (size_t)c * E);
#pragma unroll
            for (int i = 0; i < NI; i++) t2c[i].u = src[i * NT + t];
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) t23[i].h[e] = t2c[i].h[e] + q3[i].h[e];    // half add, round to nearest even
            cprev = c;
        }
        __syncthreads();                         // previous line fully scanned
        __builtin_amdgcn_s_setprio(2);           // table build + next line's loads first (see scan16.hip)
#pragma unroll
        for (int i = 0; i < NI; i++) {
            H8 t4;
#pragma unroll
            for (int e = 0; e < 4; e++) t4.h[e] = ts[i].h[e] - t2c[i].h[e];            // half subtract
            // entries 8*(i*256+t) .. +7 as dwords {T23h, T4h}: two 16-byte stores
            const uint4 a23 = t23[i].u, a4 = t4.u;
            uint4 lo, hi;
            lo.x = __builtin_amdgcn_perm(a4.x, a23.x, 0x05040100u); lo.y = __builtin_amdgcn_perm(a4.x, a23.x, 0x07060302u);
            lo.z = __builtin_amdgcn_perm(a4.y, a23.y, 0x05040100u); lo.w = __builtin_amdgcn_perm(a4.y, a23.y, 0x07060302u);
            hi.x = __builtin_amdgcn_perm(a4.z, a23.z, 0x05040100u); hi.y = __builtin_amdgcn_perm(a4.z, a23.z, 0x07060302u);
            hi.z = __builtin_amdgcn_perm(a4.w, a23.w, 0x05040100u); hi.w = __builtin_amdgcn_perm(a4.w, a23.w, 0x07060302u);
            tab[2 * (i * NT + t)] = lo;
            tab[2 * (i * NT + t) + 1] = hi;
        }
        uint4 cc = c0;
        uint32_t lb = l0;
        mcur = mnext;
        if (w + 1 < cnt) {
            load_row(mcur, ts);
            load_codes(mcur);
            mnext = mqw[12 * min(w + 2, cnt - 1) + fl];
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + off;
        const uint8_t* lp = a.lambdas + off;
        for (uint32_t j0 = (uint32_t)wave * 64; j0 < len; j0 += NT) {
            const uint32_t j = j0 + lane;
            const uint32_t jn = min(j + NT, len - 1);
            const uint4 cn = cp[jn];
            const uint32_t ln = lp[jn];
            const float l = lamtab[lb];
            float dist = __fadd_rn(__fadd_rn(b2, __fmul_rn(l, g)), __fmul_rn(__fsub_rn(__fmul_rn(l, l), l), c2));
            float tmp = 0.f;
            auto lo2f = [](uint32_t v) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(v & 0xffffu)); };
            auto hi2f = [](uint32_t v) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(v >> 16)); };
            {
                uint32_t vp[8];
                VLQ_HP_BLOCK(cc.x, cc.y, 0);
#pragma unroll
                for (int m = 0; m < 8; m++) { dist = __fadd_rn(dist, lo2f(vp[m])); tmp = __fadd_rn(tmp, hi2f(vp[m])); }
            }
            {
                uint32_t vp[8];
                VLQ_HP_BLOCK(cc.z, cc.w, 8192);
#pragma unroll
                for (int m = 0; m < 8; m++) { dist = __fadd_rn(dist, lo2f(vp[m])); tmp = __fadd_rn(tmp, hi2f(vp[m])); }
            }
            dist = __fadd_rn(dist, __fmul_rn(l, tmp));
            sel.template offer<false>(dist, pos0 + j, j < len);
            cc = cn;
            lb = ln;
        }
        total += len;
    }
    if (t == 0) cum[cnt] = total;

    ScanArgs em;                 // only the fields merge_and_emit reads
    em.k = a.k;
    em.nprobe = cnt > 0 ? cnt : 1;
    em.store_pairs = 0;
    em.ids = a.ids;
    em.D = a.D;
    em.I = a.I;
    if (cnt == 0 && t == 0) cum[1] = 0;
    merge_and_emit<KPL>(sel, smraw, cum, em, q, wave, lane, [&](int rank, int64_t& lkey, int64_t& loff) {
        const uint4 m0 = mq[3 * (int)wmap[rank]];
        lkey = (int64_t)(int32_t)m0.w;
        loff = (int64_t)(((uint64_t)m0.y << 32) | m0.x);
    });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)total);
}

// ---------------------------------------------------------------------------
// line scan for SMALL tables (M * ksub <= 2048 entries, M a multiple of 4: e.g. BASELINE configs[4] as
// written, 32 sub-quantizers x 4 bits = 512 entries): same arithmetic, same line order, same scan
// positions as the kernels above and below, organised like line16_scan_kernel -- compact line records
// grouped by anchor, the anchor's term2 row in registers, T23 rebuilt only when the anchor changes, and
// the next line's far-end row, first codes and lambda bytes requested before the current line is
// scanned.  With 512-entry tables a line costs two 2 KB rows; lines hold a handful of codes, so the
// generic kernel's per-line chain of dependent loads was what it spent its time on.
// EP = table entries per thread (E = 256 * EP).
// ---------------------------------------------------------------------------
template <int KPL, int EP>
__global__ __launch_bounds__(256) void lineS_scan_kernel(LineScanArgs a, int queue_off) {
    constexpr int NT = 256, E = NT * EP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* t23 = reinterpret_cast<float*>(smraw);                    // [E]
    float* t4 = t23 + E;                                             // [E]
    float* lamtab = t4 + E;                                          // [256]
    u64* queue = reinterpret_cast<u64*>(smraw + queue_off);          // [4][64]
    uint32_t* cum = reinterpret_cast<uint32_t*>(queue + 4 * 64);     // [w1+1] scan position of the rank-th line
    uint16_t* wmap = reinterpret_cast<uint16_t*>(cum + a.w1 + 1);    // [w1] rank -> record index

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t q = blockIdx.x;
    const int cnt = a.sel_cnt[q];
    const int Eact = a.M * a.ksub;                                   // <= E; entries beyond it are never read
    const int MW = a.M >> 2;                                         // 32-bit words per code
    const uint4* mq = reinterpret_cast<const uint4*>(a.sel_meta + q * a.w1);   // 3 x 16 bytes per record

    float m2q[EP];                        // -2 <q_m, cent_mj>, entries t, t + 256, ...
#pragma unroll
    for (int i = 0; i < EP; i++) {
        const int e = i * NT + t;
        m2q[i] = e < Eact ? __fmul_rn(-2.f, a.qtab[q * Eact + e]) : 0.f;
    }
    lamtab[t] = a.lambda_info[t];         // padded to 256 entries by the host
    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);

    float t2c[EP], ts[EP];
    uint32_t cw[8];                       // this lane's first code of the next line (M <= 32 bytes)
    uint32_t l0 = 0;
    auto load_code = [&](int64_t row, uint32_t (&w)[8]) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(a.codes + row * a.M);
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = i < MW ? p[i] : 0u;
    };
    auto prefetch = [&](const uint4 m0, const uint4 m1) __attribute__((always_inline)) {
        const int64_t off = (int64_t)(((uint64_t)m0.y << 32) | m0.x);
        const uint32_t len = m0.z;
        const int32_t s = (int32_t)m1.x;
        const float* src = a.term2 + (size_t)s * Eact;
#pragma unroll
        for (int i = 0; i < EP; i++) { const int e = i * NT + t; ts[i] = e < Eact ? src[e] : 0.f; }
        const uint32_t j = min((uint32_t)lane + 64u * wave, len - 1);
        load_code(off + j, cw);
        l0 = a.lambdas[off + j];
    };
    uint4 ma0 = make_uint4(0, 0, 0, 0), ma1 = ma0, ma2 = ma0, mb0 = ma0, mb1 = ma0, mb2 = ma0;
    if (cnt > 0) {
        ma0 = mq[0]; ma1 = mq[1]; ma2 = mq[2];
        const int w1c = min(1, cnt - 1);
        mb0 = mq[3 * w1c]; mb1 = mq[3 * w1c + 1]; mb2 = mq[3 * w1c + 2];
        prefetch(ma0, ma1);
    }
    int cprev = -1;
    uint32_t total = 0;
    for (int w = 0; w < cnt; w++) {
        const int64_t off = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane(ma0.y) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane(ma0.x));
        const uint32_t len = __builtin_amdgcn_readfirstlane(ma0.z);
        const int line = __builtin_amdgcn_readfirstlane(ma0.w);
        const float c2 = __uint_as_float(__builtin_amdgcn_readfirstlane(ma1.y));
        const float b2 = __uint_as_float(__builtin_amdgcn_readfirstlane(ma1.z));
        const float g = __uint_as_float(__builtin_amdgcn_readfirstlane(ma1.w));
        const uint32_t pos0 = __builtin_amdgcn_readfirstlane(ma2.x);
        const int rank = __builtin_amdgcn_readfirstlane(ma2.y);
        const int c = line / a.nedge;
        if (t == 0) { cum[rank] = pos0; wmap[rank] = (uint16_t)w; }
        __syncthreads();                         // previous line fully scanned
        __builtin_amdgcn_s_setprio(2);           // table build + next line's loads first (see scan16.hip)
        if (c != cprev) {                        // new anchor: its row into registers, T23 into LDS
            const float* src = a.term2 + (size_t)c * Eact;
#pragma unroll
            for (int i = 0; i < EP; i++) {
                const int e = i * NT + t;
                t2c[i] = e < Eact ? src[e] : 0.f;
                t23[e] = __fadd_rn(t2c[i], m2q[i]);
            }
            cprev = c;
        }
#pragma unroll
        for (int i = 0; i < EP; i++) t4[i * NT + t] = __fsub_rn(ts[i], t2c[i]);
        uint32_t cc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) cc[i] = cw[i];
        uint32_t lb = l0;
        ma0 = mb0; ma1 = mb1; ma2 = mb2;
        if (w + 1 < cnt) {
            prefetch(ma0, ma1);
            const int w2 = min(w + 2, cnt - 1);
            mb0 = mq[3 * w2]; mb1 = mq[3 * w2 + 1]; mb2 = mq[3 * w2 + 2];
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        for (uint32_t j0 = (uint32_t)wave * 64; j0 < len; j0 += NT) {
            const uint32_t j = j0 + lane;
            const bool valid = j < len;
            uint32_t cn[8];
            const uint32_t jn = min(j + NT, len - 1);
            if (j0 + NT < len) load_code(off + jn, cn);       // (wave-uniform) most lines fit one trip
            const uint32_t ln = j0 + NT < len ? a.lambdas[off + jn] : 0u;
            const float l = lamtab[lb];
            // PQScanMultiPassPrecomputed.cu:783-811 as written (see line16_scan_kernel)
            float dist = __fadd_rn(__fadd_rn(b2, __fmul_rn(l, g)), __fmul_rn(__fsub_rn(__fmul_rn(l, l), l), c2));
            float tmp = 0.f;
            const float* p23 = t23;
            const float* p4 = t4;
#pragma unroll
            for (int wd = 0; wd < 8; wd++) {
                if (wd < MW) {
                    const uint32_t word = cc[wd];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t code = (word >> (8 * b)) & 255u;
                        dist = __fadd_rn(dist, p23[code]);
                        tmp = __fadd_rn(tmp, p4[code]);
                        p23 += a.ksub;
                        p4 += a.ksub;
                    }
                }
            }
            dist = __fadd_rn(dist, __fmul_rn(l, tmp));
            sel.template offer<false>(dist, pos0 + j, valid);
            if (j0 + NT < len) {
#pragma unroll
                for (int i = 0; i < 8; i++) cc[i] = cn[i];
                lb = ln;
            }
        }
        total += len;
    }
    if (t == 0) cum[cnt] = total;

    ScanArgs em;                 // only the fields merge_and_emit reads
    em.k = a.k;
    em.nprobe = cnt > 0 ? cnt : 1;
    em.store_pairs = 0;
    em.ids = a.ids;
    em.D = a.D;
    em.I = a.I;
    if (cnt == 0 && t == 0) cum[1] = 0;
    merge_and_emit<KPL>(sel, smraw, cum, em, q, wave, lane, [&](int rank, int64_t& lkey, int64_t& loff) {
        const uint4 m0 = mq[3 * (int)wmap[rank]];
        lkey = (int64_t)(int32_t)m0.w;
        loff = (int64_t)(((uint64_t)m0.y << 32) | m0.x);
    });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)total);
}

template <int KPL, int EP>
static void launch_lineS_scan_t(const LineScanArgs& a, int queue_off, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(lineS_scan_kernel<KPL, EP>), smem);
    hipLaunchKernelGGL((lineS_scan_kernel<KPL, EP>), dim3((unsigned)a.nq), dim3(256), smem, s, a, queue_off);
}
template <int KPL>
static void launch_lineS_scan_e(const LineScanArgs& a, int ep, int queue_off, size_t smem, hipStream_t s) {
    if (ep <= 2) launch_lineS_scan_t<KPL, 2>(a, queue_off, smem, s);
    else if (ep <= 4) launch_lineS_scan_t<KPL, 4>(a, queue_off, smem, s);
    else launch_lineS_scan_t<KPL, 8>(a, queue_off, smem, s);
}

// ---------------------------------------------------------------------------
// line scan: one 256-thread workgroup per query walks its selected lines.  Per line
// (c, s): two LUTs in LDS,  T23 = term2[c] + (-2 <q, cent>)  and  T4 = term2[s] - term2[c],
// then per code (lambda l from the one-byte codebook)
//    dist = ((b2 + l*g) + (l*l - l)*c2) + sum T23 + l * sum T4
// (PQScanMultiPassPrecomputed.cu:744-811), at most 1024 codes per line (:728), running
// top-k by (dist, scan position) as in the IVFPQ scan.
// ---------------------------------------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void line_scan_kernel(LineScanArgs a, int lut_region) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int E = a.M * a.ksub;
    float* t23 = reinterpret_cast<float*>(smraw);                    // [E]
    float* t4 = t23 + E;                                             // [E]
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);         // [4][64]
    uint32_t* cum = reinterpret_cast<uint32_t*>(queue + 4 * 64);     // [w1+1]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t q = blockIdx.x;
    const int32_t* lq = a.sel_line + q * a.w1;
    const float* qt = a.qtab + q * E;            // <q_m, cent_mj>; T3 = -2 * this

    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);

    uint32_t pos0 = 0;
    uint64_t nscan = 0;
    for (int w = 0; w < a.w1; w++) {
        const int32_t line = lq[w];
        if (t == 0) cum[w] = pos0;
        if (line < 0) continue;
        const int64_t off = a.line_off[line];
        int64_t len = a.line_len ? a.line_len[line] : a.line_off[line + 1] - off;
        if (len > a.max_line_codes) len = a.max_line_codes;
        if (len <= 0) continue;
        const int c = line / a.nedge;
        const int s = a.edge_info[line];
        const float c2 = a.edge_dist[line];
        const float b2 = a.sel_b2[q * a.w1 + w], g = a.sel_g[q * a.w1 + w];
        const float* t2c = a.term2 + (size_t)c * E;
        const float* t2s = a.term2 + (size_t)s * E;
        __syncthreads();                         // previous line fully scanned
        __builtin_amdgcn_s_setprio(2);           // table build + next line's loads first (see scan16.hip)
        for (int e = t; e < E; e += 256) {
            const float vc = t2c[e];
            t23[e] = __fadd_rn(vc, __fmul_rn(-2.f, qt[e]));
            t4[e] = __fsub_rn(t2s[e], vc);
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        const uint8_t* cp = a.codes + off * a.M;
        const uint8_t* lp = a.lambdas + off;
        for (int64_t j0 = (int64_t)wave * 64; j0 < len; j0 += 256) {
            const int64_t j = j0 + lane;
            const bool valid = j < len;
            float dist = 0.f;
            if (valid) {
                const uint8_t* cj = cp + j * a.M;
                const float l = a.lambda_info[lp[j]];
                // the written order of PQScanMultiPassPrecomputed.cu:783-811 (see line16_scan_kernel)
                float s23 = __fadd_rn(__fadd_rn(b2, __fmul_rn(l, g)), __fmul_rn(__fsub_rn(__fmul_rn(l, l), l), c2));
                float s4 = 0.f;
                const float* p23 = t23;
                const float* p4 = t4;
                if ((a.M & 3) == 0) {
                    const uint32_t* cw = reinterpret_cast<const uint32_t*>(cj);
                    for (int wd = 0; wd < a.M / 4; wd++) {
                        const uint32_t cc = cw[wd];
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const uint32_t code = (cc >> (8 * b)) & 255u;
                            s23 = __fadd_rn(s23, p23[code]);
                            s4 = __fadd_rn(s4, p4[code]);
                            p23 += a.ksub;
                            p4 += a.ksub;
                        }
                    }
                } else {
                    for (int m = 0; m < a.M; m++) {
                        s23 = __fadd_rn(s23, p23[cj[m]]);
                        s4 = __fadd_rn(s4, p4[cj[m]]);
                        p23 += a.ksub;
                        p4 += a.ksub;
                    }
                }
                dist = __fadd_rn(s23, __fmul_rn(l, s4));
            }
            sel.offer(dist, pos0 + (uint32_t)j, valid);
        }
        pos0 += (uint32_t)len;
        nscan += (uint64_t)len;
    }
    if (t == 0) cum[a.w1] = pos0;

    ScanArgs em;                 // only the fields merge_and_emit reads
    em.k = a.k;
    em.nprobe = a.w1;
    em.store_pairs = 0;
    em.ids = a.ids;
    em.D = a.D;
    em.I = a.I;
    merge_and_emit<KPL>(sel, smraw, cum, em, q, wave, lane,
                        [&](int w, int64_t& lkey, int64_t& loff) { lkey = lq[w]; loff = a.line_off[lq[w]]; });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
}

template <int KPL>
static void launch_line_scan_t(const LineScanArgs& a, int lut_region, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(line_scan_kernel<KPL>), smem);
    hipLaunchKernelGGL((line_scan_kernel<KPL>), dim3((unsigned)a.nq), dim3(256), smem, s, a, lut_region);
}

template <int KPL>
static void launch_line16h_scan_t(const LineScanArgs& a, int queue_off, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(line16h_scan_kernel<KPL>), smem);
    hipLaunchKernelGGL((line16h_scan_kernel<KPL>), dim3((unsigned)a.nq), dim3(256), smem, s, a, queue_off);
}

template <int KPL>
static void launch_line16_scan_t(const LineScanArgs& a, int queue_off, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(line16_scan_kernel<KPL>), smem);
    hipLaunchKernelGGL((line16_scan_kernel<KPL>), dim3((unsigned)a.nq), dim3(256), smem, s, a, queue_off);
}

void launch_line_scan(const LineScanArgs& a, hipStream_t s) {
    if (a.nq <= 0) return;
    if (a.M == 16 && a.ksub == 256 && a.sel_meta && a.term2h) {
        // half tables: T23h, T4h (8 KB each) + lambda table; the merge area (4 x k keys) aliases them
        size_t lutb = (size_t)2 * 4096 * 2 + 256 * 4;
        const size_t merge = (size_t)4 * a.k * 8;
        if (lutb < merge) lutb = merge;
        const size_t smem16 = lutb + 4 * 64 * 8 + ((size_t)a.w1 + 2) * 4 + ((size_t)a.w1 + 2) * 2 + 16;
        if (a.k <= 64) launch_line16h_scan_t<1>(a, (int)lutb, smem16, s);
        else if (a.k <= 256) launch_line16h_scan_t<4>(a, (int)lutb, smem16, s);
        else launch_line16h_scan_t<16>(a, (int)lutb, smem16, s);
        return;
    }
    if (a.M == 16 && a.ksub == 256 && a.sel_meta) {
        size_t lutb = (size_t)2 * 4096 * 4 + 256 * 4;      // T23, T4, lambda table; the merge area aliases T23/T4
        const size_t smem16 = lutb + 4 * 64 * 8 + ((size_t)a.w1 + 2) * 4 + ((size_t)a.w1 + 2) * 2 + 16;
        if (a.k <= 64) launch_line16_scan_t<1>(a, (int)lutb, smem16, s);
        else if (a.k <= 256) launch_line16_scan_t<4>(a, (int)lutb, smem16, s);
        else launch_line16_scan_t<16>(a, (int)lutb, smem16, s);
        return;
    }
    if (a.sel_meta && (a.M & 3) == 0 && a.M <= 32 && a.M * a.ksub <= 2048) {
        // small tables: line records + prefetch (lineS_scan_kernel)
        const int ep = (a.M * a.ksub + 255) / 256 <= 2 ? 2 : ((a.M * a.ksub + 255) / 256 <= 4 ? 4 : 8);
        size_t lutb = (size_t)2 * 256 * ep * 4 + 256 * 4;
        const size_t merge = (size_t)4 * a.k * 8;
        if (lutb < merge) lutb = merge;
        lutb = (lutb + 15) & ~(size_t)15;
        const size_t smemS = lutb + 4 * 64 * 8 + ((size_t)a.w1 + 2) * 4 + ((size_t)a.w1 + 2) * 2 + 16;
        if (a.k <= 64) launch_lineS_scan_e<1>(a, ep, (int)lutb, smemS, s);
        else if (a.k <= 256) launch_lineS_scan_e<4>(a, ep, (int)lutb, smemS, s);
        else launch_lineS_scan_e<16>(a, ep, (int)lutb, smemS, s);
        return;
    }
    size_t lutb = (size_t)2 * a.M * a.ksub * 4;
    const size_t merge = (size_t)4 * a.k * 8;
    if (lutb < merge) lutb = merge;
    lutb = (lutb + 15) & ~(size_t)15;
    const size_t smem = lutb + 4 * 64 * 8 + ((size_t)a.w1 + 1) * 4 + 16;
    if (a.k <= 64) launch_line_scan_t<1>(a, (int)lutb, smem, s);
    else if (a.k <= 256) launch_line_scan_t<4>(a, (int)lutb, smem, s);
    else launch_line_scan_t<16>(a, (int)lutb, smem, s);
}

}  // namespace vlq
