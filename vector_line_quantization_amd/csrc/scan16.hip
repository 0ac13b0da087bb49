// List-scan kernel specialised for 16-byte codes (M = 16, ksub = 256, precomputed
// table mode 1) -- the BASELINE configs' shape.  Same arithmetic as the generic
// kernel in kernels.hip (IndexIVFPQ.cpp:631-690, :781-802), organised to keep HBM/L2
// requests in flight:
//   * probe metadata (list id, start, length, dis0) is gathered once per query into
//     LDS, so the per-probe loop has no dependent scalar global loads;
//   * term2[key] (16 KB) AND the first 16-byte code of every lane are prefetched one
//     live probe ahead; inside a list the next 256-code chunk is requested before the
//     current one is consumed;
//   * double-buffered LDS LUT, one workgroup barrier per probe;
//   * workgroups are dealt to XCDs so that queries adjacent in `qorder` (sorted by
//     nearest coarse centroid) share an L2: their term2 rows and list codes are then
//     mostly L2 hits instead of fabric reads.  Placement only affects speed.  (Walking a
//     query's probes in the spatial order of their lists as well was measured and does not
//     pay: the nearest lists must come first to tighten the admission threshold.)
#include <type_traits>

#include "kernels.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "wave_topk.cuh"

namespace vlq {

// IMI: table type 2 (multi-index: two term2 rows per list) -- a compile-time switch, the row
// addressing sits in the per-probe prefetch
template <int KPL, int NW, int NBUF, bool PIPE, bool IMI>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(((KPL == 4 || KPL == 2) && PIPE) ? 4 : 1))) void scan16_kernel(ScanArgs a, int lut_region) {
    constexpr int E = 4096;
    constexpr int NT = 64 * NW;       // threads per workgroup
    constexpr int NI = 16 / NW;       // float4 of the LUT per thread
    constexpr int QR = KPL >= 8 ? 4 : 1;   // pending-queue capacity / 64 (wave_topk.cuh)
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                         // [2][E]
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);              // [NW][64]
    ProbeMeta pm;
    pm.carve(reinterpret_cast<unsigned char*>(queue + NW * 64 * QR), a.nprobe);
    int32_t* misc = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(queue + NW * 64 * QR) +
                                               ProbeMeta::bytes(a.nprobe));    // cut, nlive
    uint16_t* ord = reinterpret_cast<uint16_t*>(misc + 2);                      // [nprobe] visited probes, in walking order
    uint32_t* wg_thr = reinterpret_cast<uint32_t*>(ord + ((a.nprobe + 1) & ~1));  // min of the waves' k-th distances

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // adc16_fixed() addresses the LUT buffers at LDS offsets 0 / 16384
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }
    uint32_t two = 2;
    asm volatile("" : "+v"(two));   // keep the shift amount in a VGPR (SDWA takes no literal)
    // XCD-aware placement: hardware deals consecutive workgroups round-robin over the 8
    // XCDs, so give XCD x the x-th contiguous chunk of the (sorted) query order.
    // small batches: a query's probes are split over a.nsplit workgroups (parts = contiguous ranges
    // of the walking order) that write partial top-k rows [part][nq][k]; merge_topk_kernel joins them
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

    // ---- per-query set-up -------------------------------------------------------
    const bool badkey = probe_meta_fill(a, q, pm, t, NT);
    float4 m2t3[NI];
    load_query_table16<NI>(a, q, t, lane, wave, m2t3);
    __syncthreads();
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
        if (lane == 0) { misc[0] = cut; misc[1] = nl; *wg_thr = f32_to_ordered(3.402823466e+38f); }
    }
    __syncthreads();
    const int nlive = misc[1];

    WaveSelect<KPL, QR, KPL >= 2> sel;   // k > 64: the merge network stays out of the scan loop's register budget
    sel.init(a.k, queue + wave * 64 * QR, lane);

    // ---- probe loop, software-pipelined one live probe ahead ----------------------
    float4 t2r[NI];
    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = make_uint4(0, 0, 0, 0);
    // the prefetched probe's metadata is carried into the iteration that scans it (wave-uniform
    // values): the loop top has no LDS round trips of its own
    uint32_t n_len = 0, n_pos0 = 0;
    float n_dis0 = 0.f;
    int64_t n_off = 0;
    auto prefetch = [&](int i) {     // i-th probe of the walking order
        if (i >= nlive) return;      // (a part may look one probe past its range: harmless loads)
        const int p = ord[i];
        const int64_t key = pm.pkey[p];
        n_len = __builtin_amdgcn_readfirstlane(pm.plen[p]);
        n_dis0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pm.pd0[p])));
        n_pos0 = __builtin_amdgcn_readfirstlane(pm.cum[p]);
        {
            const int64_t o = pm.poff[p];
            n_off = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)o >> 32)) << 32) |
                              __builtin_amdgcn_readfirstlane((uint32_t)o));
        }
        if (IMI) {
            // table type 2: sub-quantizer m = NW*i + wave takes its 1 KB slice from the row of
            // the coarse sub-index of its half (IndexIVFPQ.cpp:645-686)
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
        {   // this thread's first two codes of the list, clamped (branch-free loads)
            const uint4* cpn = reinterpret_cast<const uint4*>(a.codes) + n_off;
            const uint32_t last = n_len - 1;
            c0 = cpn[min((uint32_t)t, last)];
            c1 = cpn[min((uint32_t)t + NT, last)];
        }
    };
    const int i_begin = (int)((int64_t)part * nlive / a.nsplit), i_end = (int)((int64_t)(part + 1) * nlive / a.nsplit);
    prefetch(i_begin);
    int buf = 0;
    uint64_t nscan = 0;
    for (int i = i_begin; i < i_end; i++) {
        const uint32_t len = n_len;
        const float dis0 = n_dis0;
        const uint32_t pos0 = n_pos0;
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + n_off;
        float* L = lut + buf * E;
        if (NBUF == 1) __syncthreads();   // single LUT buffer: everyone is done scanning with it
        build_lut16<NI>(L, t, t2r, m2t3);
        uint4 cc = c0, cd = c1;
        prefetch(i + 1);
        __syncthreads();
        if (sel.dirty) {     // wave-uniform: publish this wave's k-th distance, then take the workgroup's minimum
            if (lane == 0) atomicMin(wg_thr, f32_to_ordered(sel.thr_own));
            sel.dirty = false;
        }
        sel.refresh_with(*wg_thr);
        // one copy of the list loop per LUT buffer: the buffer's LDS offset is an immediate
        auto scan_list = [&](auto bufc) {
            constexpr int B = decltype(bufc)::value;
            uint32_t j0 = (uint32_t)wave * 64;
            // two chunks per trip in four half blocks of 8 lookups: while the 8 dependent adds of one
            // half block run, the next half block's reads are in flight (counted lgkmcnt) -- LDS and
            // VALU overlap inside a wave instead of only between waves
            // (PIPE: always with one key per lane; the 32 extra registers cost the longer selections a
            // wave of occupancy -- measured k = 100 on 244-code lists: 1.27 -> 1.47 ms -- so k > 64
            // takes it only for indexes with long lists)
            for (; PIPE && j0 + NT < len; j0 += 2 * NT) {
                const uint32_t ja = j0 + lane, jb = ja + NT;
                const uint4 ca = cc, cb = cd;
                cc = cp[min(jb + NT, len - 1)];              // the next trip's two chunks
                cd = cp[min(jb + 2 * NT, len - 1)];
                float h1[8], h2[8], h3[8], h4[8];
                if (B == 0) { { float (&v)[8] = h1; VLQ_G8LO_NW(0, ca.x, ca.y); } { float (&v)[8] = h2; VLQ_G8HI_NW(0, ca.z, ca.w); } }
                else { { float (&v)[8] = h1; VLQ_G8LO_NW(16384, ca.x, ca.y); } { float (&v)[8] = h2; VLQ_G8HI_NW(16384, ca.z, ca.w); } }
                VLQ_WAIT8(8, h1);
                float da = dis0;
#pragma unroll
                for (int m = 0; m < 8; m++) da = __fadd_rn(da, h1[m]);
                asm volatile("" : "+v"(da));
                if (B == 0) { float (&v)[8] = h3; VLQ_G8LO_NW(0, cb.x, cb.y); } else { float (&v)[8] = h3; VLQ_G8LO_NW(16384, cb.x, cb.y); }
                VLQ_WAIT8(8, h2);
#pragma unroll
                for (int m = 0; m < 8; m++) da = __fadd_rn(da, h2[m]);
                asm volatile("" : "+v"(da));
                if (B == 0) { float (&v)[8] = h4; VLQ_G8HI_NW(0, cb.z, cb.w); } else { float (&v)[8] = h4; VLQ_G8HI_NW(16384, cb.z, cb.w); }
                // chunk A's admission test while chunk B's reads are in flight; the (rare) insertion itself
                // waits until nothing is in flight: no control flow between an LDS read and its wait
                const bool hit_a = __builtin_amdgcn_ballot_w64(da < sel.thr) != 0;
                VLQ_WAIT8(8, h3);
                float db = dis0;
#pragma unroll
                for (int m = 0; m < 8; m++) db = __fadd_rn(db, h3[m]);
                asm volatile("" : "+v"(db));
                VLQ_WAIT8(0, h4);
#pragma unroll
                for (int m = 0; m < 8; m++) db = __fadd_rn(db, h4[m]);
                if (hit_a) sel.offer(da, pos0 + ja, true);
                sel.offer(db, pos0 + jb, jb < len);
            }
            for (; j0 < len; j0 += NT) {
                const uint32_t j = j0 + lane;
                const uint4 cn = cp[min(j + NT, len - 1)];
                const float dis = adc16_fixed<B>(cc, dis0, two);
                sel.offer(dis, pos0 + j, j < len);
                cc = cn;
            }
        };
        if (NBUF == 1 || buf == 0) scan_list(std::integral_constant<int, 0>{});
        else scan_list(std::integral_constant<int, 1>{});
        nscan += len;
        if (NBUF == 2) buf ^= 1;
    }

    merge_and_emit<KPL, NW, QR>(sel, smraw, pm.cum, a, a.nsplit > 1 ? (int64_t)part * a.nq + q : q, wave, lane,
                        [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
    if (badkey) *a.bad_key = 1;
}

// ---------------------------------------------------------------------------
// Short-list variant (inverted multi-index and other many-list indexes: a few codes per
// list).  Building a 4096-entry LUT per probe to look up 16 x a-handful of entries is what
// the reference does (precompute_list_tables_L2 per probed list) and what bounds scan16_kernel
// there.  Here no LUT is built: the per-query part (-2 <q, cent>, 16 KB) sits in LDS once
// per query, each wave walks its own probes (no workgroup barrier in the loop), and a lane
// fetches exactly the 16 term2 entries its code addresses and forms the SAME table entries
// term2 + (-2 <q, cent>) (fvec_madd, IndexIVFPQ.cpp:641-644) before the left-to-right sum:
// identical arithmetic, identical results.
// ---------------------------------------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void scan16_short_kernel(ScanArgs a, int queue_off) {
    constexpr int E = 4096, NW = 4, NT = 256, NI = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* qtl = reinterpret_cast<float*>(smraw);                         // [16][256] -2 <q_m, cent_mj>
    u64* queue = reinterpret_cast<u64*>(smraw + queue_off);               // [NW][64]
    ProbeMeta pm;
    pm.carve(reinterpret_cast<unsigned char*>(queue + NW * 64), a.nprobe);
    int32_t* misc = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(queue + NW * 64) +
                                               ProbeMeta::bytes(a.nprobe));
    uint16_t* ord = reinterpret_cast<uint16_t*>(misc + 2);

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int64_t q;
    {
        const int64_t b = blockIdx.x;
        const int64_t s = (b & 7) * a.xcd_chunk + (b >> 3);
        if (s >= a.nq) return;
        q = a.qorder ? a.qorder[s] : s;
    }
    const int64_t* kq = a.keys + q * a.nprobe;
    const bool badkey = probe_meta_fill(a, q, pm, t, NT);
    {
        float4 m2t3[NI];
        load_query_table16<NI>(a, q, t, lane, t >> 6, m2t3);
#pragma unroll
        for (int i = 0; i < NI; i++) reinterpret_cast<float4*>(qtl)[i * NT + t] = m2t3[i];
    }
    __syncthreads();
    if (wave == 0) {
        const int cut = probe_meta_scan(a, pm, lane);
        __builtin_amdgcn_wave_barrier();
        int nl = 0;
        for (int p0 = 0; p0 < cut; p0 += 64) {
            const int p = p0 + lane;
            const bool lv = p < cut && pm.pkey[p] >= 0;
            const u64 mask = __ballot(lv);
            if (lv) ord[nl + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)p;
            nl += __popcll(mask);
        }
        if (lane == 0) { misc[0] = cut; misc[1] = nl; }
    }
    __syncthreads();
    const int nlive = misc[1];

    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);
    const uint4* codes = reinterpret_cast<const uint4*>(a.codes);
    auto first_chunk = [&](int i) {
        if (i >= nlive) return make_uint4(0, 0, 0, 0);
        const int p = ord[i];
        return codes[pm.poff[p] + min((uint32_t)lane, pm.plen[p] - 1)];
    };
    uint4 cnext = first_chunk(wave);
    for (int i = wave; i < nlive; i += NW) {          // this wave's probes, in increasing scan position
        const int p = ord[i];
        const uint32_t len = pm.plen[p];
        const float dis0 = pm.pd0[p];
        const uint32_t pos0 = pm.cum[p];
        const int64_t key = pm.pkey[p];
        const uint4* cp = codes + pm.poff[p];
        const float* row0;
        const float* row1;
        if (a.imi_nbits > 0) {      // table type 2 (IndexIVFPQ.cpp:645-686): halves from two rows
            row0 = a.term2 + (size_t)(key & ((int64_t(1) << a.imi_nbits) - 1)) * E;
            row1 = a.term2 + (size_t)(key >> a.imi_nbits) * E;
        } else {
            row0 = row1 = a.term2 + (size_t)key * E;
        }
        uint4 cc = cnext;
        cnext = first_chunk(i + NW);
        for (uint32_t j0 = 0; j0 < len; j0 += 64) {
            const uint32_t j = j0 + lane;
            const uint4 cn = cp[min(j + 64, len - 1)];
            const uint32_t w[4] = {cc.x, cc.y, cc.z, cc.w};
            float e[16];
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const uint32_t c = (w[m >> 2] >> (8 * (m & 3))) & 255u;
                const float t2 = (m < 8 ? row0 : row1)[m * 256 + c];
                e[m] = __fadd_rn(t2, qtl[m * 256 + c]);
            }
            float dis = dis0;
#pragma unroll
            for (int m = 0; m < 16; m++) dis = __fadd_rn(dis, e[m]);
            sel.offer(dis, pos0 + j, j < len);
            cc = cn;
        }
    }
    merge_and_emit<KPL, NW>(sel, smraw, pm.cum, a, q, wave, lane,
                            [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)pm.cum[a.nprobe]);
    if (badkey) *a.bad_key = 1;
}

template <int KPL>
static void launch_scan16_short_t(const ScanArgs& a, int queue_off, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan16_short_kernel<KPL>), smem);
    const unsigned grid = (unsigned)(8 * a.xcd_chunk);
    hipLaunchKernelGGL((scan16_short_kernel<KPL>), dim3(grid), dim3(256), smem, s, a, queue_off);
}

void launch_scan16_short(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    a.xcd_chunk = (int)((a.nq + 7) / 8);
    size_t region = (size_t)4096 * 4;                        // the merge area aliases the table
    const size_t merge = (size_t)4 * a.k * 8;
    if (region < merge) region = merge;
    const size_t smem = region + (size_t)4 * 64 * 8 + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 2 + 64;
    if (a.k <= 64) launch_scan16_short_t<1>(a, (int)region, smem, s);
    else if (a.k <= 256) launch_scan16_short_t<4>(a, (int)region, smem, s);
    else launch_scan16_short_t<16>(a, (int)region, smem, s);
}

template <int KPL, int NW, int NBUF, bool PIPE, bool IMI>
static void launch_scan16_i(const ScanArgs& a, int lut_region, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan16_kernel<KPL, NW, NBUF, PIPE, IMI>), smem);
    const unsigned grid = (unsigned)(8 * a.xcd_chunk);
    hipLaunchKernelGGL((scan16_kernel<KPL, NW, NBUF, PIPE, IMI>), dim3(grid), dim3(64 * NW), smem, s, a, lut_region);
}
template <int KPL, int NW, int NBUF, bool PIPE>
static void launch_scan16_t(const ScanArgs& a, int lut_region, size_t smem, hipStream_t s) {
    if (a.imi_nbits > 0) launch_scan16_i<KPL, NW, NBUF, PIPE, true>(a, lut_region, smem, s);
    else launch_scan16_i<KPL, NW, NBUF, PIPE, false>(a, lut_region, smem, s);
}

void launch_scan16(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    if (a.nsplit < 1) a.nsplit = 1;
    a.xcd_chunk = (int)((a.nq * a.nsplit + 7) / 8);
    // k <= 64: 8 waves per workgroup share one LUT (32 waves per CU at 4 workgroups);
    // larger k keeps more selection state per wave, so stay at 4 waves
    // Measured alternatives (r01, MI355X, bench data): 8 waves per workgroup 0.95 ms, single
    // LUT buffer with 6 workgroups per CU 0.78 ms, two probes of lookahead 0.82 ms, two
    // queries per workgroup sharing term2 rows 0.93 ms -- against 0.78-0.82 ms for this
    // configuration (4 waves, double-buffered LUT, one probe of lookahead).
    const int nw = 4;
    size_t lutb = (size_t)2 * 4096 * 4;
    const size_t merge = (size_t)nw * a.k * 8;
    if (lutb < merge) lutb = merge;
    const size_t tail = (size_t)nw * 64 * 8 * (a.k > 256 ? 4 : 1) + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 2 + 8 + 64;
    const size_t smem = lutb + tail;
#ifdef VLQ_EXPERIMENTS
    if (getenv("VLQ_FORCE_KPL4")) {     // the k > 64 kernel on a small k: separates code structure from insertion statistics
        if (a.long_lists) launch_scan16_t<4, 4, 2, true>(a, (int)lutb, smem, s);
        else launch_scan16_t<4, 4, 2, false>(a, (int)lutb, smem, s);
        return;
    }
#endif
    if (a.k <= 64) launch_scan16_t<1, 4, 2, true>(a, (int)lutb, smem, s);
    else if (a.k <= 128) {          // recall@100: half the merge network of the 256-key list
        if (a.long_lists) launch_scan16_t<2, 4, 2, true>(a, (int)lutb, smem, s);
        else launch_scan16_t<2, 4, 2, false>(a, (int)lutb, smem, s);
    } else if (a.k <= 256) {
        if (a.long_lists) launch_scan16_t<4, 4, 2, true>(a, (int)lutb, smem, s);
        else launch_scan16_t<4, 4, 2, false>(a, (int)lutb, smem, s);
    } else if (a.k <= 512) launch_scan16_t<8, 4, 2, false>(a, (int)lutb, smem, s);
    else launch_scan16_t<16, 4, 2, false>(a, (int)lutb, smem, s);
}

// ---------------------------------------------------------------------------
// query order: counting sort of the queries by their nearest coarse centroid
// (keys[q][0]).  Order inside a bin is arbitrary -- it only influences which
// workgroups run next to each other, never a result.
// ---------------------------------------------------------------------------
__global__ void qorder_hist_kernel(const int64_t* __restrict__ keys, int64_t nq, int nprobe,
                                   int nlist, int* __restrict__ hist, const int* __restrict__ list_rank,
                                   int shift, int nbins) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int64_t k0 = keys[q * nprobe];
    const bool ok = k0 >= 0 && k0 < nlist;
    atomicAdd(&hist[ok ? ((list_rank ? list_rank[k0] : (int)k0) >> shift) : nbins - 1], 1);
}

// prefix of the bin counts (every workgroup recomputes it in LDS: at most 16 Ki bins) + placement
__global__ __launch_bounds__(256) void qorder_place_kernel(const int64_t* __restrict__ keys, int64_t nq, int nprobe,
                                                           int nlist, const int* __restrict__ hist,
                                                           int* __restrict__ cnt, int* __restrict__ qorder,
                                                           const int* __restrict__ list_rank, int shift, int nbins) {
    extern __shared__ int pre[];                 // [nbins] exclusive prefix
    __shared__ int part[256];
    const int t = threadIdx.x;
    const int per = (nbins + 255) / 256;
    const int b0 = t * per;
    int sum = 0;
    for (int i = 0; i < per; i++) if (b0 + i < nbins) sum += hist[b0 + i];
    part[t] = sum;
    __syncthreads();
    for (int sft = 1; sft < 256; sft <<= 1) {
        const int v = t >= sft ? part[t - sft] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = 0; i < per; i++)
        if (b0 + i < nbins) { pre[b0 + i] = run; run += hist[b0 + i]; }
    __syncthreads();
    const int64_t q = (int64_t)blockIdx.x * 256 + t;
    if (q >= nq) return;
    const int64_t k0 = keys[q * nprobe];
    const bool ok = k0 >= 0 && k0 < nlist;
    const int bin = ok ? ((list_rank ? list_rank[k0] : (int)k0) >> shift) : nbins - 1;
    qorder[pre[bin] + atomicAdd(&cnt[bin], 1)] = (int)q;
}

void launch_query_order(const int64_t* keys, int64_t nq, int nprobe, int nlist, int* hist,
                        int* qorder, hipStream_t s, const int* list_rank) {
    if (nq <= 0) return;
    // at most 16 Ki bins (the prefix is recomputed per workgroup in LDS): many-list indexes are binned
    // by the high bits of the list id / rank -- for a multi-index key that is its second sub-index
    int shift = 0;
    while (((int64_t)nlist >> shift) > 16384) shift++;
    const int nbins = (int)(((int64_t)nlist - 1) >> shift) + 2;        // last bin: invalid keys
    const size_t stride = query_order_bins_padded(nlist);               // hist | cnt, one aligned memset
    (void)hipMemsetAsync(hist, 0, 2 * stride * sizeof(int), s);
    const unsigned g = (unsigned)((nq + 255) / 256);
    hipLaunchKernelGGL(qorder_hist_kernel, dim3(g), dim3(256), 0, s, keys, nq, nprobe, nlist, hist, list_rank,
                       shift, nbins);
    const size_t smem = (size_t)nbins * sizeof(int);
    ensure_dynamic_lds(reinterpret_cast<const void*>(qorder_place_kernel), smem);
    hipLaunchKernelGGL(qorder_place_kernel, dim3(g), dim3(256), smem, s, keys, nq, nprobe, nlist, hist,
                       hist + stride, qorder, list_rank, shift, nbins);
}

}  // namespace vlq
