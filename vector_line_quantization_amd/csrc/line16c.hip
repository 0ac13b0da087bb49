// VLQ scan for 16-byte codes with the QUERY-INDEPENDENT half of the distance hoisted out of the search.
//
// The reference's scan (pqScanPrecomputedMultiPassGraph, gpu/impl/PQScanMultiPassPrecomputed.cu:783-811) forms per
// code   dist = term1 + la*term6 + (la*la - la)*term5;  dist += term23[m][code_m] (m ascending);
//        tmp  = 0 + term4[m][code_m] (m ascending);     out = dist + la*tmp
// with term4 = term2[s] - term2[c] (loadPrecomputedTermGraph :313-334).  `tmp` depends on the code's bytes and on the
// line (c, s) the code is stored on -- not on the query -- and a stored vector lives on exactly one line.  So
//        p = la * tmp
// is a constant of the stored code.  line_consts_kernel computes it once per database state with exactly the operations
// the scan kernels in line.hip perform per query (one __fsub_rn per entry, __fadd_rn from 0 left to right, one
// __fmul_rn; float16 tables: half subtract, float adds); the scan then adds it with the same final __fadd_rn.  Results are
// bit-identical to line16_scan_kernel / line16h_scan_kernel and to oracle/vlq_oracle.cpp; what disappears from the
// search is the far-end row term2[s] (1024 of the 1088 rows a query read at the reference driver's geometry), the T4
// table and half of the LDS gathers.  Cost: 4 bytes per stored vector (per table precision in use).
//
// What is left is the structure of the plain IVFPQ scan: one table T23 = term2[c] + (-2<q, .>) per ANCHOR centroid
// (at most nprobe per query), built once for all the kept lines of that anchor.  The kept lines arrive grouped by anchor
// (LineMeta records, line_select_kernel); their codes are scanned as ONE virtual stream per anchor: lane t of trip n
// takes virtual code n*NT + t and finds its line by walking the per-query prefix sums in LDS -- lines hold ~240 codes
// at the driver's geometry, a line per trip (the old organisation) left 256-wide trips mostly empty.  Code, lambda
// byte and constant of a virtual position are requested PF trips ahead of their use, across anchor boundaries.
#include <cstdlib>

#include "line.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "wave_topk.cuh"

namespace vlq {

// ---------------------------------------------------------------------------
// per-code constants: one workgroup per line
// ---------------------------------------------------------------------------
template <bool HALF>
__global__ __launch_bounds__(256) void line_consts_kernel(const uint8_t* __restrict__ codes, const uint8_t* __restrict__ lambdas,
                                                          const int64_t* __restrict__ line_off, const int64_t* __restrict__ line_len,
                                                          const int32_t* __restrict__ edge_info, const float* __restrict__ term2,
                                                          const uint16_t* __restrict__ term2h, const float* __restrict__ lambda_info,
                                                          int nedge, int M, int ksub, int64_t nlines, float* __restrict__ out,
                                                          const int* __restrict__ newcnt) {
    const int64_t line = blockIdx.x;
    if (line >= nlines) return;
    const int64_t off = line_off[line];
    const int64_t len = line_len ? line_len[line] : line_off[line + 1] - off;
    if (len <= 0) return;
    // newcnt (after an append in place): only the newcnt[line] vectors the line has just received, at its end
    const int64_t first = newcnt ? len - (int64_t)newcnt[line] : 0;
    if (first >= len) return;
    const size_t E = (size_t)M * ksub;
    const size_t rc = (size_t)(line / nedge) * E, rs = (size_t)edge_info[line] * E;
    for (int64_t j = first + threadIdx.x; j < len; j += 256) {
        const uint8_t* cj = codes + (off + j) * M;
        const float l = lambda_info[lambdas[off + j]];
        float tmp = 0.f;
        for (int m = 0; m < M; m++) {
            const size_t idx = (size_t)m * ksub + cj[m];
            if (HALF) {
                const _Float16 hs = __builtin_bit_cast(_Float16, term2h[rs + idx]);
                const _Float16 hc = __builtin_bit_cast(_Float16, term2h[rc + idx]);
                const _Float16 t4 = hs - hc;                                  // half subtract (:313-334, Math<Half8>::sub)
                tmp = __fadd_rn(tmp, (float)t4);
            } else {
                tmp = __fadd_rn(tmp, __fsub_rn(term2[rs + idx], term2[rc + idx]));
            }
        }
        out[off + j] = __fmul_rn(l, tmp);
    }
}

void launch_line_consts(const uint8_t* codes, const uint8_t* lambdas, const int64_t* line_off, const int64_t* line_len,
                        const int32_t* edge_info, const float* term2, const uint16_t* term2h, const float* lambda_info,
                        int nedge, int M, int ksub, int64_t nlines, float* out, hipStream_t s, const int* newcnt) {
    if (nlines <= 0) return;
    // (the grid's x dimension holds 2^31 - 1 workgroups; the host checks nlist * nedge < 2^31)
    if (term2h)
        hipLaunchKernelGGL(line_consts_kernel<true>, dim3((unsigned)nlines), dim3(256), 0, s, codes, lambdas, line_off, line_len,
                           edge_info, term2, term2h, lambda_info, nedge, M, ksub, nlines, out, newcnt);
    else
        hipLaunchKernelGGL(line_consts_kernel<false>, dim3((unsigned)nlines), dim3(256), 0, s, codes, lambdas, line_off, line_len,
                           edge_info, term2, term2h, lambda_info, nedge, M, ksub, nlines, out, newcnt);
}

// ---------------------------------------------------------------------------
// the scan
// ---------------------------------------------------------------------------
struct L16cLayout {
    int queue_off;    // [NW][64] u64 pending queues (behind the table / the merge area)
    int meta_off;     // [w1] uint4 {offv lo, offv hi, posv, vend} then [w1] float2 {g, c2}
    int grp_off;      // [gcap+1] u32 first virtual position, [gcap] i32 anchor, [gcap] f32 b2, [NW + 2] scratch
    int gcap;         // groups (anchors) a query can have: min(w1, nprobe)
    int nparts;       // workgroups per query: part p scans virtual positions [V*p/nparts, V*(p+1)/nparts) and, if there is
                      // more than one, writes its k best (distance, scan position) keys for line16c_merge_kernel
};

typedef _Float16 h16x2c __attribute__((ext_vector_type(2)));
union H8c { uint4 u; h16x2c h[4]; };

template <int KPL, int NW, int PF, bool HALF>
__global__ __launch_bounds__(64 * NW) void line16c_scan_kernel(LineScanArgs a, L16cLayout lay) {
    constexpr int E = 4096, NT = 64 * NW, NI = 16 / NW, NH = 8 / NW, RPT = 1024 / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* t23 = reinterpret_cast<float*>(smraw);                          // [E] at LDS byte 0 (adc16_halves<0>)
    float* lamtab = t23 + E;                                               // [256]
    u64* queue = reinterpret_cast<u64*>(smraw + lay.queue_off);            // [NW][64]
    uint4* metaA = reinterpret_cast<uint4*>(smraw + lay.meta_off);         // [w1]
    float2* metaB = reinterpret_cast<float2*>(metaA + a.w1);               // [w1]
    uint32_t* gv = reinterpret_cast<uint32_t*>(smraw + lay.grp_off);       // [gcap+1]
    int32_t* gc = reinterpret_cast<int32_t*>(gv + lay.gcap + 1);           // [gcap]
    float* gb2 = reinterpret_cast<float*>(gc + lay.gcap);                  // [gcap]
    uint32_t* wsum = reinterpret_cast<uint32_t*>(gb2 + lay.gcap);          // [NW] + {ng, V}

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (__builtin_amdgcn_groupstaticsize() != 0) return;                   // the gather offsets are absolute
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
#ifdef VLQ_L16C_TIMING
    const uint64_t tk0 = wall_clock64();
#endif
    const int64_t q = blockIdx.x / lay.nparts;
    const int part = (int)(blockIdx.x - q * lay.nparts);
    const int cnt = min(a.sel_cnt[q], a.w1);
    const uint4* mq = reinterpret_cast<const uint4*>(a.sel_meta + q * a.w1);   // 3 x 16 bytes per record

    static_assert(NW == 4 || NW == 8, "a row of 4096 halves is NH x NT 16-byte words");
    // -2 <q_m, cent_mj>: fp32 entries 4*(i*NT+t) .. +3 (i < NI), or halves 8*(i*NT+t) .. +7 (i < NH)
    float4 m2q[NI];
    H8c q3[NH];
    if constexpr (HALF) {
        const uint4* qt = reinterpret_cast<const uint4*>(a.qtabh + q * E);
#pragma unroll
        for (int i = 0; i < NH; i++) q3[i].u = qt[i * NT + t];
    } else {
        const float4* qt = reinterpret_cast<const float4*>(a.qtab + q * E);
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const float4 v = qt[i * NT + t];
            m2q[i] = make_float4(__fmul_rn(-2.f, v.x), __fmul_rn(-2.f, v.y), __fmul_rn(-2.f, v.z), __fmul_rn(-2.f, v.w));
        }
    }
    if (t < 256) lamtab[t] = a.lambda_info[t];         // padded to 256 entries by the host

    // ---- per-query line directory: prefix sums of the kept lines' lengths in record order (= grouped by anchor) ----
    int ng = 0;
    uint32_t V = 0;
    {
        uint32_t len_r[RPT], head_r[RPT];
        uint32_t local = 0;                            // lens | heads << 21  (lens <= 2^20, heads <= 1024)
        int prevc = -1;
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            const int w = t * RPT + i;
            len_r[i] = 0; head_r[i] = 0;
            if (w < cnt) {
                const uint4 m0 = mq[3 * w];
                const int c = (int)mq[3 * w + 2].z;                  // LineMeta::anchor
                if (i == 0) prevc = w > 0 ? (int)mq[3 * (w - 1) + 2].z : -1;
                len_r[i] = m0.z;
                head_r[i] = c != prevc ? 1u : 0u;
                prevc = c;
                local += len_r[i] + (head_r[i] << 21);
            }
        }
        uint32_t incl = local;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t o = __shfl_up(incl, sft, 64);
            if (lane >= sft) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) { const uint32_t s = wsum[w]; if (w < wave) base += s; tot += s; }
        ng = (int)(tot >> 21);
        V = tot & ((1u << 21) - 1u);
        uint32_t run = base + incl - local;
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            const int w = t * RPT + i;
            if (w < cnt) {
                const uint4 m0 = mq[3 * w], m1 = mq[3 * w + 1], m2 = mq[3 * w + 2];
                const uint32_t vbeg = run & ((1u << 21) - 1u);
                const int gi = (int)(run >> 21);                     // heads before this record
                const int64_t off = (int64_t)(((uint64_t)m0.y << 32) | m0.x);
                const int64_t ov = off - (int64_t)vbeg;
                metaA[w] = make_uint4((uint32_t)ov, (uint32_t)((uint64_t)ov >> 32), m2.x - vbeg, vbeg + len_r[i]);
                metaB[w] = make_float2(__uint_as_float(m1.w), __uint_as_float(m1.y));     // g, c2
                if (head_r[i] && gi < lay.gcap) { gv[gi] = vbeg; gc[gi] = (int)m2.z; gb2[gi] = __uint_as_float(m1.z); }
                run += len_r[i] + (head_r[i] << 21);
            }
        }
        if (ng > lay.gcap) ng = lay.gcap;              // (cannot happen: one group per probed centroid)
        if (t == 0) gv[ng] = V;
    }
    WaveSelect<KPL, 1, (KPL >= 2)> sel;
    sel.init(a.k, queue + wave * 64, lane);
    __syncthreads();

    // this workgroup's share of the query's virtual positions
    const uint32_t lo = (uint32_t)((uint64_t)V * (uint32_t)part / (uint32_t)lay.nparts);
    const uint32_t hi = (uint32_t)((uint64_t)V * (uint32_t)(part + 1) / (uint32_t)lay.nparts);
    int g0 = 0;                                        // first group with positions >= lo (wave-uniform walk; <= gcap steps)
    if (lo < hi) while (g0 + 1 < ng && gv[g0 + 1] <= lo) g0++;
    g0 = __builtin_amdgcn_readfirstlane(g0);

    // ---- the stream of (code, lambda, constant) requests, PF trips ahead of the adds ----
    uint4 s_code[PF];
    uint32_t s_lam[PF], s_pos[PF];
    float s_pc[PF], s_g[PF], s_c2[PF];
    bool s_ok[PF];
    int pg = g0;                                       // group of the next trip to request (wave-uniform)
    uint32_t pv0 = lo;                                 // its first virtual position ...
    uint32_t pghi = lo < hi ? min(gv[g0 + 1], hi) : lo;   // ... and the end of its group (== pv0: nothing left)
    // line lookup, per lane: a pointer into the directory that walks forward with the lane's positions (a line holds ~240
    // codes at the driver's geometry, a trip advances NT positions: one or two steps per trip).  A wave-uniform variant (one
    // lane reads the entry when the wave's 64 positions share a line, SGPR broadcast) measured 2.5 % slower: the directory
    // reads are not what the kernel waits for (HBM is, see launch_line16c_scan).
    int pline = 0;
    uint32_t pvend = 0;
    if (lo < hi) {                                     // record holding position lo: first w with vend[w] > lo
        int a0 = 0, b0 = cnt - 1;
        while (a0 < b0) { const int mid = (a0 + b0) >> 1; if (metaA[mid].w <= lo) a0 = mid + 1; else b0 = mid; }
        pline = a0;
        pvend = metaA[a0].w;
    }
    const uint4* codes16 = reinterpret_cast<const uint4*>(a.codes);
    auto request = [&](uint4& code, uint32_t& lam, float& pc, float& g, float& c2, uint32_t& pos, bool& ok) __attribute__((always_inline)) {
        const uint32_t pv = pv0 + (uint32_t)t;
        ok = pv < pghi;
        while (ok && pv >= pvend) { pline++; pvend = metaA[pline].w; }
        const uint4 A = metaA[ok ? pline : 0];
        const float2 B = metaB[ok ? pline : 0];
        const int64_t ov = (int64_t)(((uint64_t)A.y << 32) | A.x);
        const int64_t ci = ok ? ov + (int64_t)pv : 0;
        code = codes16[ci];
        lam = a.lambdas[ci];
        pc = a.pconst[ci];
        pos = A.z + pv;
        g = B.x;
        c2 = B.y;
        // next trip: same group, or the first trip of the next one (groups are contiguous in the virtual order)
        pv0 += NT;
        if (pv0 >= pghi) {
            pv0 = pghi;
            if (pghi < hi) { pg++; pghi = __builtin_amdgcn_readfirstlane(min(gv[pg + 1], hi)); }
        }
    };
    float4 t2r[NI];
    H8c t2h[NH];
    auto load_row = [&](int c) __attribute__((always_inline)) {
        if constexpr (HALF) {
            const uint4* src = reinterpret_cast<const uint4*>(a.term2h + (size_t)c * E);
#pragma unroll
            for (int i = 0; i < NH; i++) t2h[i].u = src[i * NT + t];
        } else {
            const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)c * E);
#pragma unroll
            for (int i = 0; i < NI; i++) t2r[i] = src[i * NT + t];
        }
    };
#ifdef VLQ_L16C_TIMING
    const uint64_t tk1 = wall_clock64();
#endif
    if (lo < hi) {
        load_row(__builtin_amdgcn_readfirstlane(gc[g0]));
#pragma unroll
        for (int s = 0; s < PF; s++) request(s_code[s], s_lam[s], s_pc[s], s_g[s], s_c2[s], s_pos[s], s_ok[s]);
    }

    for (int g = g0; lo < hi && g < ng && gv[g] < hi; g++) {
        __syncthreads();                               // everyone is done with the previous anchor's table
        __builtin_amdgcn_s_setprio(2);                 // table build + the next row's loads first (see scan16.hip)
        if constexpr (HALF) {
            // T23h = hadd(term2h[c], term3h) in half (PQScanMultiPassPrecomputed.cu:54-75), widened exactly to float
#pragma unroll
            for (int i = 0; i < NH; i++) {
                float4 lo, hi;
                const h16x2c s0 = t2h[i].h[0] + q3[i].h[0], s1 = t2h[i].h[1] + q3[i].h[1];
                const h16x2c s2 = t2h[i].h[2] + q3[i].h[2], s3 = t2h[i].h[3] + q3[i].h[3];
                lo = make_float4((float)s0.x, (float)s0.y, (float)s1.x, (float)s1.y);
                hi = make_float4((float)s2.x, (float)s2.y, (float)s3.x, (float)s3.y);
                const int e8 = i * NT + t;
                reinterpret_cast<float4*>(t23)[2 * e8] = lo;
                reinterpret_cast<float4*>(t23)[2 * e8 + 1] = hi;
            }
        } else {
            build_lut16<NI>(t23, t, t2r, m2q);
        }
        if (g + 1 < ng && gv[g + 1] < hi) load_row(__builtin_amdgcn_readfirstlane(gc[g + 1]));
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        const float b2 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(gb2[g])));
        const uint32_t glen = __builtin_amdgcn_readfirstlane(min(gv[g + 1], hi) - max(gv[g], lo));
        for (uint32_t j0 = 0; j0 < glen; j0 += NT) {
            const uint4 cc = s_code[0];
            const uint32_t lb = s_lam[0], pos = s_pos[0];
            const float pc = s_pc[0], lg = s_g[0], c2 = s_c2[0];
            const bool ok = s_ok[0];
#pragma unroll
            for (int s = 0; s + 1 < PF; s++) {
                s_code[s] = s_code[s + 1]; s_lam[s] = s_lam[s + 1]; s_pos[s] = s_pos[s + 1];
                s_pc[s] = s_pc[s + 1]; s_g[s] = s_g[s + 1]; s_c2[s] = s_c2[s + 1]; s_ok[s] = s_ok[s + 1];
            }
            request(s_code[PF - 1], s_lam[PF - 1], s_pc[PF - 1], s_g[PF - 1], s_c2[PF - 1], s_pos[PF - 1], s_ok[PF - 1]);
            const float l = lamtab[lb];
            // PQScanMultiPassPrecomputed.cu:783-811 as written: dist = term1 + la*term6 + (la*la-la)*term5, then
            // dist += term23[m] for m ascending; the result is dist + la*tmp with la*tmp = the stored constant
            float dist = __fadd_rn(__fadd_rn(b2, __fmul_rn(l, lg)), __fmul_rn(__fsub_rn(__fmul_rn(l, l), l), c2));
            dist = adc16_halves<0>(cc, dist, two);
            dist = __fadd_rn(dist, pc);
            // positions do not arrive in increasing order (records are grouped by anchor): the full
            // (distance, position) key decides among equal distances
            sel.template offer<false>(dist, pos, ok);
        }
    }

#ifdef VLQ_L16C_TIMING
    const uint64_t tk2 = wall_clock64();
    uint64_t tbuild = 0;
#endif
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)(hi - lo));
    if (lay.nparts > 1) {
        // raw keys out: scan positions are global to the query, so line16c_merge_kernel orders the parts' candidates
        // exactly as one workgroup scanning every position would have
        if (merge_waves<KPL, NW>(sel, smraw, a.k, wave, lane)) {
            u64* out = a.part_keys + ((size_t)q * lay.nparts + part) * a.k;
#pragma unroll
            for (int r = 0; r < KPL; r++) {
                const int e = r * 64 + lane;
                if (e < a.k) out[e] = sel.best[r];
            }
        }
        return;
    }
    // ---- scan position -> (line, offset) tables for the rows out; they take the line directory's place ----
    __syncthreads();
    uint32_t* cum = reinterpret_cast<uint32_t*>(smraw + lay.meta_off);     // [w1+1] scan position of the rank-th line
    uint16_t* wmap = reinterpret_cast<uint16_t*>(cum + a.w1 + 2);          // [w1] rank -> record index
    for (int w = t; w < cnt; w += NT) {
        const uint4 m2 = mq[3 * w + 2];
        cum[m2.y] = m2.x;
        wmap[m2.y] = (uint16_t)w;
    }
    if (t == 0) { cum[cnt] = V; if (cnt == 0) cum[1] = 0; }

    ScanArgs em;                 // only the fields merge_and_emit reads
    em.k = a.k;
    em.nprobe = cnt > 0 ? cnt : 1;
    em.store_pairs = 0;
    em.ids = a.ids;
    em.D = a.D;
    em.I = a.I;
    merge_and_emit<KPL, NW>(sel, smraw, cum, em, q, wave, lane, [&](int rank, int64_t& lkey, int64_t& loff) {
        const uint4 m0 = mq[3 * (int)wmap[rank]];
        lkey = (int64_t)(int32_t)m0.w;
        loff = (int64_t)(((uint64_t)m0.y << 32) | m0.x);
    });
#ifdef VLQ_L16C_TIMING
    if (t == 0) {
        const uint64_t tk3 = wall_clock64();
        atomicAdd(a.ncode + 2, (unsigned long long)(tk1 - tk0));
        atomicAdd(a.ncode + 3, (unsigned long long)(tk2 - tk1));
        atomicAdd(a.ncode + 4, (unsigned long long)(tk3 - tk2));
        atomicAdd(a.ncode + 5, 1ull);
    }
#endif
}

// joins the parts of a query: one wave per query selects the k smallest of the parts' keys -- a total order (distance, scan
// position), so the rows are what one workgroup scanning every position returns -- and translates positions to ids
template <int KPL>
__global__ __launch_bounds__(64) void line16c_merge_kernel(LineScanArgs a, int nparts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    u64* queue = reinterpret_cast<u64*>(smraw);                            // [64]
    uint32_t* cum = reinterpret_cast<uint32_t*>(queue + 64);               // [w1+2]
    uint16_t* wmap = reinterpret_cast<uint16_t*>(cum + a.w1 + 2);          // [w1]
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int cnt = min(a.sel_cnt[q], a.w1);
    const uint4* mq = reinterpret_cast<const uint4*>(a.sel_meta + q * a.w1);
    uint32_t V = 0;
    for (int w = lane; w < cnt; w += 64) {
        const uint4 m0 = mq[3 * w], m2 = mq[3 * w + 2];
        cum[m2.y] = m2.x;
        wmap[m2.y] = (uint16_t)w;
        V += m0.z;
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) V += __shfl_xor(V, sft, 64);
    if (lane == 0) { cum[cnt] = V; if (cnt == 0) cum[1] = 0; }
    WaveSelect<KPL> sel;
    sel.init(a.k, queue, lane);
    for (int p = 0; p < nparts; p++) {
        const u64* src = a.part_keys + ((size_t)q * nparts + p) * a.k;
        for (int e0 = 0; e0 < a.k; e0 += 64) {
            const int e = e0 + lane;
            const bool valid = e < a.k;
            sel.offer_key(valid ? src[e] : kMaxKey, valid);
        }
    }
    sel.flush();
    __builtin_amdgcn_wave_barrier();
    ScanArgs em;
    em.k = a.k;
    em.nprobe = cnt > 0 ? cnt : 1;
    em.store_pairs = 0;
    em.ids = a.ids;
    em.D = a.D;
    em.I = a.I;
    emit_rows<KPL>(sel, cum, em, q, lane, [&](int rank, int64_t& lkey, int64_t& loff) {
        const uint4 m0 = mq[3 * (int)wmap[rank]];
        lkey = (int64_t)(int32_t)m0.w;
        loff = (int64_t)(((uint64_t)m0.y << 32) | m0.x);
    });
}

template <int KPL, int NW, int PF, bool HALF>
static void launch_line16c_t(const LineScanArgs& a, const L16cLayout& lay, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(line16c_scan_kernel<KPL, NW, PF, HALF>), smem);
    hipLaunchKernelGGL((line16c_scan_kernel<KPL, NW, PF, HALF>), dim3((unsigned)(a.nq * lay.nparts)), dim3(64 * NW), smem, s, a, lay);
    if (lay.nparts > 1) {
        const size_t sm = 64 * 8 + ((size_t)a.w1 + 2) * 4 + (size_t)a.w1 * 2 + 16;
        hipLaunchKernelGGL((line16c_merge_kernel<KPL>), dim3((unsigned)a.nq), dim3(64), sm, s, a, lay.nparts);
    }
}

// workgroups per query.  A chip full of workgroups is bound by HBM (launch_line16c_scan), not by the idle slots of its last
// round: measured at the driver's geometry with 2000 queries (768 resident workgroups), 1 / 2 / 3 / 4 / 6 parts per query:
// 2.23 / 2.47 / 2.53 / 2.86 / 2.94 ms -- every part pays its own directory build (12 us), merge (27 us) and first table.
// So a batch is split only while it cannot fill the chip: 250 queries (a 2000-query batch sharded over 8 GPUs)
// 0.488 -> 0.355 ms with 3 parts.
int line16c_parts(int64_t nq, int k, int max_parts) {
    static const int cus = [] {        // (once per process: the query is not cheap, and this runs per search page)
        int dev = 0, n = 256;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
            n = pr.multiProcessorCount;
        return n;
    }();
    const int64_t slots = (int64_t)cus * (k <= 256 ? 3 : 2);
    int p = 1;
    while (p < max_parts && nq * p * 10 < slots * 9) p++;
    return p;
}

template <int NW, int PF, bool HALF>
static void launch_line16c_k(const LineScanArgs& a, hipStream_t s) {
    L16cLayout lay;
    size_t region = (size_t)4096 * 4 + 256 * 4;                 // table + lambda table; the merge area aliases them
    const size_t merge = (size_t)NW * a.k * 8;
    if (region < merge) region = merge;
    region = (region + 15) & ~(size_t)15;
    lay.queue_off = (int)region;
    lay.meta_off = (int)(region + (size_t)NW * 64 * 8);
    const size_t meta = (size_t)a.w1 * 24;                      // >= (w1 + 2) * 4 + w1 * 2 for the tables that replace it
    lay.gcap = a.nprobe > 0 && a.nprobe < a.w1 ? a.nprobe : a.w1;
    lay.grp_off = (int)((lay.meta_off + meta + 16 + 15) & ~(size_t)15);
    lay.nparts = a.part_keys && a.nparts > 1 ? a.nparts : 1;
    const size_t smem = (size_t)lay.grp_off + ((size_t)lay.gcap + 1) * 4 + (size_t)lay.gcap * 8 + (NW + 2) * 4 + 16;
    if (a.k <= 64) launch_line16c_t<1, NW, PF, HALF>(a, lay, smem, s);
    else if (a.k <= 128) launch_line16c_t<2, NW, PF, HALF>(a, lay, smem, s);
    else if (a.k <= 256) launch_line16c_t<4, NW, PF, HALF>(a, lay, smem, s);
    else launch_line16c_t<16, NW, PF, HALF>(a, lay, smem, s);
}

bool line16c_supports(const LineScanArgs& a) {
    return a.M == 16 && a.ksub == 256 && a.sel_meta && a.pconst && a.w1 <= 1024;
}

void launch_line16c_scan(const LineScanArgs& a, hipStream_t s) {
    if (a.nq <= 0) return;
    // 8 waves share a table: 24 waves per CU at 3 workgroups (77 VGPRs); measured at the reference driver's geometry (1 B
    // codes, 2000 queries): 4 waves / 2 trips ahead 2.78 ms, 4 / 3 2.88, 4 / 1 2.73, 4 / 4 2.95, 8 / 3 2.51, 8 / 2 2.28
    if (a.term2h) launch_line16c_k<8, 2, true>(a, s);
    else launch_line16c_k<8, 2, false>(a, s);
}

}  // namespace vlq
