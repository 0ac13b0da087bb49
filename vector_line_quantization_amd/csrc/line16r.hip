// VLQ line scan for 16-byte codes that RECOMPUTES the term-2 rows instead of reading them.
//
// line16_scan_kernel (line.hip) reads one 16 KB row term2[s] per kept line: at the reference driver's
// geometry (65 536 centroids x 64 edges, w1 = 1024) that is 35.6 GB of rows against 8.3 GB of codes per
// 2000-query batch, 44.8 GB at 6.3 TB/s = the achievable HBM rate (profiles/r03_pmc_vlq.txt): the kernel
// sat on the memory roofline of bytes the algorithm does not need.  A row is a pure function of the
// far-end centroid (96 floats) and the PQ codebook:
//     term2[x][m][j] = |cent_mj|^2 + 2 <x_m, cent_mj>                 (impl/IVFPQ.cu:599-684)
// so every thread keeps ITS 16 codebook entries (16 x dsub floats) and their norms in registers for the
// whole query and rebuilds its 16 entries of a row from the centroid's sub-vectors, which arrive in
// scalar registers (they are wave-uniform: entry e = 4*(i*256+t)+r belongs to sub-quantizer
// m = 4*i + wave).  The operations and their order are those of pq_tables_kernel mode 2 (kernels.hip) --
// ip_sse_order, then rnorm + 2*ip, unfused -- so the rebuilt row is bit-identical to the stored one and
// every distance is bit-identical to line16_scan_kernel's and the oracle's.
//
// With the rows gone the scan is bound by its LDS gathers, so the two tables are stored INTERLEAVED,
// {T23[m][j], T4[m][j]} as one 8-byte entry: one ds_read_b64 per code byte serves both look-ups (a
// ds_read_b64 costs the LDS what a ds_read_b32 costs).  The table is double-buffered (2 x 32 KB, the
// buffer offset rides in the ds_read immediate): line w+1's table is written while other waves still
// scan line w, one workgroup barrier per line.
//
// Arithmetic per code as written in PQScanMultiPassPrecomputed.cu:783-811 (see line16_scan_kernel).
#include "line.h"
#include "scan_common.cuh"
#include "sse_order.cuh"
#include "wave_topk.cuh"

namespace vlq {

// 8 code bytes (words W0, W1) -> 8 x {T23, T4}: interleaved table at LDS byte O + m*2048 (m = 0..7 of
// this half).  One SDWA op per byte makes byte*8; buffer and sub-quantizer offsets are immediates.
#define VLQ_R16_BLOCK(W0, W1, O)                                                                                   \
    asm volatile(                                                                                                  \
        "v_lshlrev_b32_sdwa %8, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"   \
        "v_lshlrev_b32_sdwa %9, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"   \
        "v_lshlrev_b32_sdwa %10, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"  \
        "v_lshlrev_b32_sdwa %11, %18, %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"  \
        "v_lshlrev_b32_sdwa %12, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"  \
        "v_lshlrev_b32_sdwa %13, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"  \
        "v_lshlrev_b32_sdwa %14, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"  \
        "v_lshlrev_b32_sdwa %15, %18, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"  \
        "ds_read_b64 %0, %8 offset:" #O "+0\n\t"                                                                   \
        "ds_read_b64 %1, %9 offset:" #O "+2048\n\t"                                                                \
        "ds_read_b64 %2, %10 offset:" #O "+4096\n\t"                                                               \
        "ds_read_b64 %3, %11 offset:" #O "+6144\n\t"                                                               \
        "ds_read_b64 %4, %12 offset:" #O "+8192\n\t"                                                               \
        "ds_read_b64 %5, %13 offset:" #O "+10240\n\t"                                                              \
        "ds_read_b64 %6, %14 offset:" #O "+12288\n\t"                                                              \
        "ds_read_b64 %7, %15 offset:" #O "+14336"                                                                  \
        : "=&v"(pr[0]), "=&v"(pr[1]), "=&v"(pr[2]), "=&v"(pr[3]), "=&v"(pr[4]), "=&v"(pr[5]), "=&v"(pr[6]),        \
          "=&v"(pr[7]), "=&v"(ad[0]), "=&v"(ad[1]), "=&v"(ad[2]), "=&v"(ad[3]), "=&v"(ad[4]), "=&v"(ad[5]),        \
          "=&v"(ad[6]), "=&v"(ad[7])                                                                               \
        : "v"(W0), "v"(W1), "v"(three)                                                                             \
        : "memory")

// the reads of VLQ_R16_BLOCK have landed: every use of pr[] is ordered behind this
#define VLQ_R16_WAIT()                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                            \
                 : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]),      \
                   "+v"(pr[7])                                                                                     \
                 :                                                                                                 \
                 : "memory")

typedef float f32x2 __attribute__((ext_vector_type(2)));

// <x, y> over DSUB components in the order of fvec_inner_product (utils.cpp:509-533, sse_order.cuh): four lane
// sums s_l = x_l y_l + x_{l+4} y_{l+4} + ..., then (s0 + s1) + (s2 + s3), products and sums unfused.
// ip_sse_order() also performs the SSE code's additions of zero (0 + p at the start, s + 0 for the padded tail);
// they are dropped here.  x + 0 and 0 + x return x for every x != 0, so the two can differ only in the SIGN of a
// zero result, and the only use of the result, rn + 2 * ip with rn = |cent|^2 >= +0, is the same float for
// ip = +0 and ip = -0.  The table entries are therefore bit-identical to pq_tables_kernel's.
template <int DSUB>
__device__ __forceinline__ float ip_lanes(const float (&x)[DSUB], const float (&y)[DSUB]) {
    static_assert(DSUB >= 4, "four SSE lanes");
    float s[4];
#pragma unroll
    for (int l = 0; l < 4; l++) s[l] = __fmul_rn(x[l], y[l]);
#pragma unroll
    for (int i = 4; i < DSUB; i++) s[i & 3] = __fadd_rn(s[i & 3], __fmul_rn(x[i], y[i]));
    return __fadd_rn(__fadd_rn(s[0], s[1]), __fadd_rn(s[2], s[3]));
}

// NW waves per workgroup; a thread owns NI * 4 table entries: e = 4 * (i * NT + t) + r, i < NI, r < 4, i.e.
// sub-quantizer m = i * NW + wave (wave-uniform) and centroids j = 4 * lane + r.
//   NW = 4 (the instantiated shape): 16 entries per thread, 16 x dsub codebook floats in registers, 2 waves per SIMD.
//   NW = 8 / 16 (8 / 4 entries per thread, 4 waves per SIMD; a line's codes are scanned by ONE group of four
//   waves, the NW/4 groups taking the lines in turn) were measured and are slower: every wave pays the per-line
//   fixed cost (record fields, barrier) and the 8-wave shape spills -- 12.1 / 11.8 ms against 8.8 ms at C5.
template <int KPL, int DSUB, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void line16r_scan_kernel(LineScanArgs a, int queue_off) {
    constexpr int E = 4096, NT = 64 * NW, NI = 16 / NW, NG = NW / 4, D = 16 * DSUB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    // LDS: [2 x 32 KB interleaved tables at byte 0][lambda table 1 KB][queue NW x 64 keys][cum][wmap]
    float* lamtab = reinterpret_cast<float*>(smraw + 65536);         // [256]
    u64* queue = reinterpret_cast<u64*>(smraw + queue_off);          // [NW][64]
    uint32_t* cum = reinterpret_cast<uint32_t*>(queue + NW * 64);    // [w1+1] scan position of the rank-th line
    uint16_t* wmap = reinterpret_cast<uint16_t*>(cum + a.w1 + 1);    // [w1] rank -> record index

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2, gw = wave & 3;                        // scan group, wave inside the group
    if (__builtin_amdgcn_groupstaticsize() != 0) return;             // the table offsets in the gathers are absolute
    uint32_t three = 3;
    asm volatile("" : "+v"(three));
    const int64_t q = blockIdx.x;
    const int cnt = a.sel_cnt[q];
    // a line record is 12 dwords: lane l < 12 holds dword l, fields are read with v_readlane
    const uint32_t* mqw = reinterpret_cast<const uint32_t*>(a.sel_meta + q * a.w1);
    const int fl = lane < 12 ? lane : 0;

    // this thread's codebook entries and their norms
    float cent[NI][4][DSUB], rn[NI][4];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int m = i * NW + wave;
        const float* cp = a.pq_cent + ((size_t)m * 256 + 4 * lane) * DSUB;     // 4 * DSUB contiguous floats
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < DSUB; c++) cent[i][r][c] = cp[r * DSUB + c];
        const float4 n4 = *reinterpret_cast<const float4*>(a.pq_rnorm + m * 256 + 4 * lane);
        rn[i][0] = n4.x; rn[i][1] = n4.y; rn[i][2] = n4.z; rn[i][3] = n4.w;
    }
    if (t < 256) lamtab[t] = a.lambda_info[t];         // padded to 256 entries by the host
    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);

    // sub-vectors m = i * NW + wave of centroid x: wave-uniform addresses of memory that is constant for the
    // whole launch, read through the constant address space so that they are scalar loads into SGPRs
    typedef const __attribute__((address_space(4))) float* cfp;
    auto load_sub = [&](int x, float (&xs)[NI][DSUB]) __attribute__((always_inline)) {
        cfp src = (cfp)(uintptr_t)(a.coarse + (size_t)x * D + wave * DSUB);
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int c = 0; c < DSUB; c++) xs[i][c] = src[i * NW * DSUB + c];
    };
    // the thread's entries of the term-2 row of a centroid: rn + 2 <x_m, cent_mj>, pq_tables_kernel mode 2
    auto rebuild = [&](const float (&xs)[NI][DSUB], float (&row)[NI][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                row[i][r] = __fadd_rn(rn[i][r], __fmul_rn(2.f, ip_lanes<DSUB>(xs[i], cent[i][r])));   // fvec_madd(r_norms, 2.0, tab)
    };

    float t2c[NI][4], t23[NI][4];
    uint32_t mcur = 0, mnext = 0;
    uint4 c0 = make_uint4(0, 0, 0, 0);
    uint32_t l0 = 0;
    float xs[NI][DSUB];                   // far-end sub-vectors of the line about to be tabled
    auto rec_off = [&](uint32_t rec) __attribute__((always_inline)) {
        return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(rec, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane(rec, 0));
    };
    // first code chunk of a line for the group that will scan it (rec: lane-distributed record)
    auto prefetch_codes = [&](uint32_t rec) __attribute__((always_inline)) {
        const int64_t off = rec_off(rec);
        const uint32_t len = __builtin_amdgcn_readlane(rec, 2);
        const uint32_t j = min((uint32_t)(gw * 64 + lane), len - 1);
        c0 = reinterpret_cast<const uint4*>(a.codes)[off + j];
        l0 = a.lambdas[off + j];
    };
    if (cnt > 0) {
        mcur = mqw[fl];
        mnext = mqw[12 * min(1, cnt - 1) + fl];
        load_sub(__builtin_amdgcn_readlane(mcur, 4), xs);
        if (grp == 0) prefetch_codes(mcur);
    }
    int cprev = -1;
    uint32_t total = 0;

    // one line: its table into buffer BUF, barrier, scan by the line's group
    auto do_line = [&](int w, auto bufc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        const int64_t off = rec_off(mcur);
        const uint32_t len = __builtin_amdgcn_readlane(mcur, 2);
        const int line = __builtin_amdgcn_readlane(mcur, 3);
        const float c2 = __uint_as_float(__builtin_amdgcn_readlane(mcur, 5));
        const float b2 = __uint_as_float(__builtin_amdgcn_readlane(mcur, 6));
        const float g = __uint_as_float(__builtin_amdgcn_readlane(mcur, 7));
        const uint32_t pos0 = __builtin_amdgcn_readlane(mcur, 8);
        const int rank = __builtin_amdgcn_readlane(mcur, 9);
        const int c = line / a.nedge;
        const bool mine = (w % NG) == grp;       // this wave's group scans line w
        if (t == 0) { cum[rank] = pos0; wmap[rank] = (uint16_t)w; }
        if (c != cprev) {                        // new anchor: its row, and T23 = term2[c] + (-2 <q, .>)
            float xc[NI][DSUB];
            load_sub(c, xc);
            rebuild(xc, t2c);
            const float4* qt = reinterpret_cast<const float4*>(a.qtab + q * E);
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const float4 v = qt[i * NT + t];
                t23[i][0] = __fadd_rn(t2c[i][0], __fmul_rn(-2.f, v.x));
                t23[i][1] = __fadd_rn(t2c[i][1], __fmul_rn(-2.f, v.y));
                t23[i][2] = __fadd_rn(t2c[i][2], __fmul_rn(-2.f, v.z));
                t23[i][3] = __fadd_rn(t2c[i][3], __fmul_rn(-2.f, v.w));
            }
            cprev = c;
        }
        {
            float t2s[NI][4];
            rebuild(xs, t2s);
            float4* tab = reinterpret_cast<float4*>(smraw + BUF * 32768);
#pragma unroll
            for (int i = 0; i < NI; i++) {       // T4 = term2[s] - term2[c]
                tab[2 * (i * NT + t)] = make_float4(t23[i][0], __fsub_rn(t2s[i][0], t2c[i][0]),
                                                    t23[i][1], __fsub_rn(t2s[i][1], t2c[i][1]));
                tab[2 * (i * NT + t) + 1] = make_float4(t23[i][2], __fsub_rn(t2s[i][2], t2c[i][2]),
                                                        t23[i][3], __fsub_rn(t2s[i][3], t2c[i][3]));
            }
        }
        uint4 cc = c0;
        uint32_t lb = l0;
        mcur = mnext;
        if (w + 1 < cnt) {
            load_sub(__builtin_amdgcn_readlane(mcur, 4), xs);
            if (((w + 1) % NG) == grp) prefetch_codes(mcur);
            mnext = mqw[12 * min(w + 2, cnt - 1) + fl];
        }
        __syncthreads();                         // table of line w complete; every wave is done with line w-2's buffer
        if (mine) {
            const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + off;
            const uint8_t* lp = a.lambdas + off;
            for (uint32_t j0 = (uint32_t)gw * 64; j0 < len; j0 += 256) {
                const uint32_t j = j0 + lane;
                const uint32_t jn = min(j + 256, len - 1);
                const uint4 cn = cp[jn];
                const uint32_t ln = lp[jn];
                const float l = lamtab[lb];
                float dist = __fadd_rn(__fadd_rn(b2, __fmul_rn(l, g)), __fmul_rn(__fsub_rn(__fmul_rn(l, l), l), c2));
                float tmp = 0.f;
                {
                    f32x2 pr[8];
                    uint32_t ad[8];
                    if (BUF == 0) VLQ_R16_BLOCK(cc.x, cc.y, 0); else VLQ_R16_BLOCK(cc.x, cc.y, 32768);
                    VLQ_R16_WAIT();
#pragma unroll
                    for (int m = 0; m < 8; m++) { dist = __fadd_rn(dist, pr[m].x); tmp = __fadd_rn(tmp, pr[m].y); }
                }
                {
                    f32x2 pr[8];
                    uint32_t ad[8];
                    if (BUF == 0) VLQ_R16_BLOCK(cc.z, cc.w, 16384); else VLQ_R16_BLOCK(cc.z, cc.w, 49152);
                    VLQ_R16_WAIT();
#pragma unroll
                    for (int m = 0; m < 8; m++) { dist = __fadd_rn(dist, pr[m].x); tmp = __fadd_rn(tmp, pr[m].y); }
                }
                dist = __fadd_rn(dist, __fmul_rn(l, tmp));
                // positions do not arrive in increasing order (records are grouped by anchor, lines are spread
                // over the groups): the full (distance, position) key decides among equal distances
                sel.template offer<false>(dist, pos0 + j, j < len);
                cc = cn;
                lb = ln;
            }
        }
        total += len;
    };
    for (int w = 0; w < cnt; w += 2) {
        do_line(w, std::integral_constant<int, 0>());
        if (w + 1 < cnt) do_line(w + 1, std::integral_constant<int, 1>());
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
    merge_and_emit<KPL, NW>(sel, smraw, cum, em, q, wave, lane, [&](int rank, int64_t& lkey, int64_t& loff) {
        const uint32_t* rec = mqw + 12 * (int)wmap[rank];
        lkey = (int64_t)(int32_t)rec[3];
        loff = (int64_t)(((uint64_t)rec[1] << 32) | rec[0]);
    });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)total);
}

template <int KPL, int DSUB, int NW>
static void launch_line16r_t(const LineScanArgs& a, hipStream_t s) {
    const size_t lutb = 65536 + 1024;      // two interleaved tables + lambda table; the merge area aliases the tables
    const size_t smem = lutb + (size_t)NW * 64 * 8 + ((size_t)a.w1 + 2) * 4 + ((size_t)a.w1 + 2) * 2 + 16;
    ensure_dynamic_lds(reinterpret_cast<const void*>(line16r_scan_kernel<KPL, DSUB, NW>), smem);
    hipLaunchKernelGGL((line16r_scan_kernel<KPL, DSUB, NW>), dim3((unsigned)a.nq), dim3(64 * NW), smem, s, a, (int)lutb);
}

bool line16r_supports(const LineScanArgs& a, int dsub) {
    return a.M == 16 && a.ksub == 256 && a.sel_meta && a.coarse && a.pq_cent && a.pq_rnorm &&
           (dsub == 4 || dsub == 6 || dsub == 8) && a.k <= 256;
}

void launch_line16r_scan(const LineScanArgs& a, int dsub, hipStream_t s) {
    if (a.nq <= 0) return;
#define VLQ_R16_DISPATCH(K)                                   \
    do {                                                      \
        if (dsub == 4) launch_line16r_t<K, 4, 4>(a, s);       \
        else if (dsub == 6) launch_line16r_t<K, 6, 4>(a, s);  \
        else launch_line16r_t<K, 8, 4>(a, s);                 \
    } while (0)
    if (a.k <= 64) VLQ_R16_DISPATCH(1);
    else VLQ_R16_DISPATCH(4);
#undef VLQ_R16_DISPATCH
}

}  // namespace vlq
