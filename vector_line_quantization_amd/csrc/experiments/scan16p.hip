// Two-queries-per-workgroup list scan for 16-byte codes (M = 16, ksub = 256, table mode 1/2).
//
// Same arithmetic as scan16.hip (IndexIVFPQ.cpp:631-690, :781-802).  What it adds: the
// workgroup serves the two queries that are neighbours in the nearest-centroid order
// (launch_query_order) and walks the UNION of their probe lists.  A list both queries
// visit (78 % of the scanned codes on the bench data) is scanned ONCE against a LUT that
// interleaves the two queries' entries: one ds_read_b64 per code byte returns both
// queries' table values -- the LDS cost and the address arithmetic of one query's
// lookup -- and the two left-to-right sums advance side by side.  Lists only one of the
// two visits take the single-query path.  Each query keeps its own running selection
// with its own scan positions (position = place in ITS probe order), so distances, ties
// and labels are exactly those of the one-query kernels: the selection key (distance,
// position) is a total order and does not care in which order lists are walked.
#include <type_traits>

#include "kernels.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "wave_topk.cuh"

namespace vlq {

typedef float f2 __attribute__((ext_vector_type(2)));

// Pair form of VLQ_G16_ASM: the LUT holds {query a, query b} interleaved (8 bytes per entry,
// sub-quantizer stride 2048 B, at LDS offset 0), one ds_read_b64 fetches both queries' entries
// of one code byte: 2 LDS cycles per 32 lanes, the same as a ds_read_b32 (MI355X LDS table).
// Two blocks of 8 lookups (sub-quantizers 0-7, 8-15) keep the kernel within 128 VGPRs.
#define VLQ_G8P_LO(W0, W1)                                                                   \
    asm volatile(                                                                          \
        "v_lshlrev_b32_sdwa %8, %14, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %9, %14, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %10, %14, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %11, %14, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b64 %0, %8 offset:0\n\t" \
        "ds_read_b64 %1, %9 offset:2048\n\t" \
        "ds_read_b64 %2, %10 offset:4096\n\t" \
        "ds_read_b64 %3, %11 offset:6144\n\t" \
        "v_lshlrev_b32_sdwa %8, %14, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %9, %14, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %10, %14, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %11, %14, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b64 %4, %8 offset:8192\n\t" \
        "ds_read_b64 %5, %9 offset:10240\n\t" \
        "ds_read_b64 %6, %10 offset:12288\n\t" \
        "ds_read_b64 %7, %11 offset:14336\n\t" \
        "s_waitcnt lgkmcnt(0)"                                                                                 \
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&v"(ta0), "=&v"(ta1), "=&v"(ta2), "=&v"(ta3)                                                                               \
        : "v"(W0), "v"(W1), "v"(three)                                                     \
        : "memory")
#define VLQ_G8P_HI(W0, W1)                                                                   \
    asm volatile(                                                                          \
        "v_lshlrev_b32_sdwa %8, %14, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %9, %14, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %10, %14, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %11, %14, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b64 %0, %8 offset:16384\n\t" \
        "ds_read_b64 %1, %9 offset:18432\n\t" \
        "ds_read_b64 %2, %10 offset:20480\n\t" \
        "ds_read_b64 %3, %11 offset:22528\n\t" \
        "v_lshlrev_b32_sdwa %8, %14, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t" \
        "v_lshlrev_b32_sdwa %9, %14, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
        "v_lshlrev_b32_sdwa %10, %14, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t" \
        "v_lshlrev_b32_sdwa %11, %14, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
        "ds_read_b64 %4, %8 offset:24576\n\t" \
        "ds_read_b64 %5, %9 offset:26624\n\t" \
        "ds_read_b64 %6, %10 offset:28672\n\t" \
        "ds_read_b64 %7, %11 offset:30720\n\t" \
        "s_waitcnt lgkmcnt(0)"                                                                                 \
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&v"(ta0), "=&v"(ta1), "=&v"(ta2), "=&v"(ta3)                                                                               \
        : "v"(W0), "v"(W1), "v"(three)                                                     \
        : "memory")
// both queries' distances of one code: disa/disb = dis0 + tab[0][c0] + ... + tab[15][c15],
// strictly left to right (IndexIVFPQ.cpp:788-794)
__device__ __forceinline__ void adc16_pair(const uint4 cc, float& disa, float& disb, uint32_t three) {
    uint32_t ta0, ta1, ta2, ta3;
    {
        f2 p[8];
        VLQ_G8P_LO(cc.x, cc.y);
#pragma unroll
        for (int m = 0; m < 8; m++) { disa = __fadd_rn(disa, p[m].x); disb = __fadd_rn(disb, p[m].y); }
    }
    {
        f2 p[8];
        VLQ_G8P_HI(cc.z, cc.w);
#pragma unroll
        for (int m = 0; m < 8; m++) { disa = __fadd_rn(disa, p[m].x); disb = __fadd_rn(disb, p[m].y); }
    }
}

#ifndef VLQ_P_OCC
#define VLQ_P_OCC 4
#endif
template <int KPL>
__global__ __launch_bounds__(256, VLQ_P_OCC) void scan16p_kernel(ScanArgs a, int meta_bytes) {
    constexpr int E = 4096;
    constexpr int NW = 4, NT = 256, NI = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                              // 32 KB: pair LUT / single LUT / merge area
    u64* queue = reinterpret_cast<u64*>(smraw + 2 * E * 4);                    // [2][NW][64]
    unsigned char* tail = reinterpret_cast<unsigned char*>(queue + 2 * NW * 64);
    ProbeMeta pma, pmb;
    pma.carve(tail, a.nprobe);
    pmb.carve(tail + meta_bytes, a.nprobe);
    int16_t* partner = reinterpret_cast<int16_t*>(tail + 2 * meta_bytes);      // [nprobe] b-probe sharing a's list, or -1
    int16_t* bmatched = partner + a.nprobe;                                    // [nprobe]
    int16_t* sch_a = bmatched + a.nprobe;                                      // [2*nprobe] schedule: probe of a or -1
    int16_t* sch_b = sch_a + 2 * a.nprobe;                                     //                      probe of b or -1
    int32_t* misc = reinterpret_cast<int32_t*>(sch_b + 2 * a.nprobe);          // cut_a, cut_b, nsched, dup

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }   // LUT must sit at LDS address 0
    uint32_t two = 2, three = 3;
    asm volatile("" : "+v"(two), "+v"(three));
    // XCD-aware placement of PAIR slots (see scan16.hip)
    int64_t qa, qb;
    bool hasb;
    {
        const int64_t b = blockIdx.x;
        const int64_t s = (b & 7) * a.xcd_chunk + (b >> 3);
        const int64_t npair = (a.nq + 1) >> 1;
        if (s >= npair) return;
        hasb = 2 * s + 1 < a.nq;
        qa = a.qorder ? a.qorder[2 * s] : 2 * s;
        qb = hasb ? (a.qorder ? a.qorder[2 * s + 1] : 2 * s + 1) : qa;
    }

    // ---- per-pair set-up -----------------------------------------------------------
    bool badkey = probe_meta_fill(a, qa, pma, t, NT);
    badkey |= probe_meta_fill(a, qb, pmb, t, NT);
    float4 qta[NI], qtb[NI];
    load_query_table16<NI>(a, qa, t, lane, wave, qta);
    load_query_table16<NI>(a, qb, t, lane, wave, qtb);
    if (t < 4) misc[t] = 0;
    __syncthreads();
    if (wave == 0) { const int cut = probe_meta_scan(a, pma, lane); if (lane == 0) misc[0] = cut; }
    if (wave == 1) { const int cut = probe_meta_scan(a, pmb, lane); if (lane == 0) misc[1] = hasb ? cut : 0; }
    __syncthreads();
    const int cuta = misc[0], cutb = misc[1];
    // which of b's probes visits the same list as a's probe p?  (keys of one query are distinct
    // when they come from the coarse stage; a caller's duplicates switch pairing off)
    for (int p = t; p < a.nprobe; p += NT) bmatched[p] = 0;
    __syncthreads();
    for (int p = t; p < cuta; p += NT) {
        const int32_t key = pma.pkey[p];
        int pb = -1;
        bool dup = false;
        if (key >= 0) {
            for (int j = 0; j < cutb; j++)
                if (pmb.pkey[j] == key) { if (pb < 0) pb = j; else dup = true; }
            for (int j = 0; j < p; j++) if (pma.pkey[j] == key) dup = true;
        }
        partner[p] = (int16_t)pb;
        if (pb >= 0) bmatched[pb] = 1;
        if (dup) misc[3] = 1;
    }
    __syncthreads();
    if (wave == 0) {
#ifdef VLQ_P_NOPAIR
        const bool nopair = true;
#else
        const bool nopair = misc[3] != 0;
#endif
        int n = 0;
        for (int p0 = 0; p0 < cuta; p0 += 64) {
            const int p = p0 + lane;
            const bool lv = p < cuta && pma.pkey[p] >= 0;
            const u64 mask = __ballot(lv);
            if (lv) {
                const int slot = n + __popcll(mask & ((1ull << lane) - 1ull));
                sch_a[slot] = (int16_t)p;
                sch_b[slot] = nopair ? (int16_t)-1 : partner[p];
            }
            n += __popcll(mask);
        }
        for (int p0 = 0; p0 < cutb; p0 += 64) {
            const int p = p0 + lane;
            const bool lv = p < cutb && pmb.pkey[p] >= 0 && (nopair || !bmatched[p]);
            const u64 mask = __ballot(lv);
            if (lv) {
                const int slot = n + __popcll(mask & ((1ull << lane) - 1ull));
                sch_a[slot] = (int16_t)-1;
                sch_b[slot] = (int16_t)p;
            }
            n += __popcll(mask);
        }
        if (lane == 0) misc[2] = n;
    }
    __syncthreads();
    const int nsched = misc[2];

    WaveSelect<KPL> sela, selb;
    sela.init(a.k, queue + wave * 64, lane);
    selb.init(a.k, queue + (NW + wave) * 64, lane);

    // ---- schedule loop, one entry of lookahead for term2 row and first code chunk ------
    float4 t2r[NI];
    uint4 c0 = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](int i) {
        if (i >= nsched) return;
        const int pa = sch_a[i], pb = sch_b[i];
        ProbeMeta pm;
        pm.carve(tail + (pa >= 0 ? 0 : meta_bytes), a.nprobe);
        const int p = pa >= 0 ? pa : pb;
        const int64_t key = pm.pkey[p];
        if (a.imi_nbits > 0) {     // table type 2 (IndexIVFPQ.cpp:645-686)
            const int64_t ki0 = key & ((int64_t(1) << a.imi_nbits) - 1), ki1 = key >> a.imi_nbits;
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) {
                const int64_t ki = (NW * i2 + wave) < 8 ? ki0 : ki1;
                t2r[i2] = reinterpret_cast<const float4*>(a.term2 + (size_t)ki * E)[i2 * NT + t];
            }
        } else {
            const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)key * E);
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) t2r[i2] = src[i2 * NT + t];
        }
        c0 = reinterpret_cast<const uint4*>(a.codes)[pm.poff[p] + min((uint32_t)t, pm.plen[p] - 1)];
    };
    prefetch(0);
    for (int i = 0; i < nsched; i++) {
        const int pa = sch_a[i], pb = sch_b[i];
        ProbeMeta pm;
        pm.carve(tail + (pa >= 0 ? 0 : meta_bytes), a.nprobe);
        const int p = pa >= 0 ? pa : pb;
        const uint32_t len = pm.plen[p];
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + pm.poff[p];
        uint4 cc = c0;
        if (pa >= 0 && pb >= 0) {
            // ---- list visited by both: interleaved LUT, entry e at byte 8*e = {a, b} ----
            const float d0a = pma.pd0[pa], d0b = pmb.pd0[pb];
            const uint32_t posa = pma.cum[pa], posb = pmb.cum[pb];
            float4* L4 = reinterpret_cast<float4*>(lut);
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) {
                float4 sa, sb;   // sim_table = term2[key] + (-2) * sim_table_2 (fvec_madd, IndexIVFPQ.cpp:641-644)
                sa.x = __fadd_rn(t2r[i2].x, qta[i2].x); sb.x = __fadd_rn(t2r[i2].x, qtb[i2].x);
                sa.y = __fadd_rn(t2r[i2].y, qta[i2].y); sb.y = __fadd_rn(t2r[i2].y, qtb[i2].y);
                sa.z = __fadd_rn(t2r[i2].z, qta[i2].z); sb.z = __fadd_rn(t2r[i2].z, qtb[i2].z);
                sa.w = __fadd_rn(t2r[i2].w, qta[i2].w); sb.w = __fadd_rn(t2r[i2].w, qtb[i2].w);
                L4[(i2 * NT + t) * 2] = make_float4(sa.x, sb.x, sa.y, sb.y);
                L4[(i2 * NT + t) * 2 + 1] = make_float4(sa.z, sb.z, sa.w, sb.w);
            }
            prefetch(i + 1);
            __syncthreads();
#pragma unroll 2
            for (uint32_t j0 = (uint32_t)wave * 64; j0 < len; j0 += NT) {
                const uint32_t j = j0 + lane;
                const uint4 cn = cp[min(j + NT, len - 1)];
                float disa = d0a, disb = d0b;
                adc16_pair(cc, disa, disb, three);
                sela.offer(disa, posa + j, j < len);
                selb.template offer<false>(disb, posb + j, j < len);   // b's positions are not visited in order
                cc = cn;
            }
        } else {
            // ---- list visited by one of the two: the single-query path of scan16.hip ----
            const float dis0 = pm.pd0[p];
            const uint32_t pos0 = pm.cum[p];
            if (pa >= 0) build_lut16<NI>(lut, t, t2r, qta); else build_lut16<NI>(lut, t, t2r, qtb);
            prefetch(i + 1);
            __syncthreads();
            auto scan_single = [&](WaveSelect<KPL>& sel) {
#pragma unroll 2
                for (uint32_t j0 = (uint32_t)wave * 64; j0 < len; j0 += NT) {
                    const uint32_t j = j0 + lane;
                    const uint4 cn = cp[min(j + NT, len - 1)];
                    const float dis = adc16_fixed<0>(cc, dis0, two);
                    sel.template offer<false>(dis, pos0 + j, j < len);
                    cc = cn;
                }
            };
            if (pa >= 0) scan_single(sela); else scan_single(selb);
        }
        __syncthreads();     // single LUT buffer: everyone is done with it
    }

    // ---- merge the four waves' selections: wave 0 finishes query a, wave 1 query b -------
    sela.flush();
    selb.flush();
    u64* mb = reinterpret_cast<u64*>(smraw);   // [2][NW][k], aliases the LUT (free after the loop's last barrier)
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = r * 64 + lane;
        if (e < a.k) { mb[wave * a.k + e] = sela.best[r]; mb[(NW + wave) * a.k + e] = selb.best[r]; }
    }
    __syncthreads();
    auto finish = [&](WaveSelect<KPL>& sel, int which, const ProbeMeta& pm, int64_t q) {
        for (int w = 0; w < NW; w++) {
            if (w == wave) continue;
            for (int e0 = 0; e0 < a.k; e0 += 64) {
                const int e = e0 + lane;
                const bool valid = e < a.k;
                const u64 key = valid ? mb[(which * NW + w) * a.k + e] : kMaxKey;
                sel.offer_key(key, valid);
            }
        }
        sel.flush();
        const int64_t* kq = a.keys + q * a.nprobe;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const int e = r * 64 + lane;
            if (e >= a.k) continue;
            const u64 key = sel.best[r];
            float dis = 3.402823466e+38f;          // Heap.h:318-321 padding
            int64_t id = -1;
            if (key != kMaxKey) {
                dis = ordered_to_f32((uint32_t)(key >> 32));
                const uint32_t pos = (uint32_t)key;
                int lo = 0, hi = a.nprobe;         // last probe p with cum[p] <= pos
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (pm.cum[mid] <= pos) lo = mid; else hi = mid;
                }
                const int64_t o = pos - pm.cum[lo];
                id = a.store_pairs ? (kq[lo] << 32 | o) : a.ids[pm.poff[lo] + o];   // IndexIVFPQ.cpp:798
            }
            a.D[q * a.k + e] = dis;
            a.I[q * a.k + e] = id;
        }
    };
    if (wave == 0) finish(sela, 0, pma, qa);
    if (wave == 1 && hasb) finish(selb, 1, pmb, qb);
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)pma.cum[a.nprobe] + (hasb ? pmb.cum[a.nprobe] : 0u));
    if (badkey) *a.bad_key = 1;
}

template <int KPL>
static void launch_scan16p_t(const ScanArgs& a, int meta_bytes, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan16p_kernel<KPL>), smem);
    const unsigned grid = (unsigned)(8 * a.xcd_chunk);
    hipLaunchKernelGGL((scan16p_kernel<KPL>), dim3(grid), dim3(256), smem, s, a, meta_bytes);
}

bool scan16p_supports(const ScanArgs& a) { return a.k <= 256 && a.nprobe <= 64 && a.nq >= 2; }

void launch_scan16p(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    const int64_t npair = (a.nq + 1) / 2;
    a.xcd_chunk = (int)((npair + 7) / 8);
    const int meta_bytes = (int)(((size_t)a.nprobe * 24 + 8 + 15) & ~(size_t)15);
    const size_t smem = (size_t)2 * 4096 * 4 + (size_t)2 * 4 * 64 * 8 + (size_t)2 * meta_bytes +
                        (size_t)a.nprobe * 2 * 6 + 32 + 64;
    if (a.k <= 64) launch_scan16p_t<1>(a, meta_bytes, smem, s);
    else launch_scan16p_t<4>(a, meta_bytes, smem, s);
}

}  // namespace vlq
