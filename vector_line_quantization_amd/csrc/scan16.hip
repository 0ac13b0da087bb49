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
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "walk_order.cuh"
#include "wave_topk.cuh"

namespace vlq {

// Phase clocks of the query-major kernel (diagnostic builds only: make FLAGS_scan16=-DVLQ_SCAN16_PHASES; tools/scan16_phases.py).
// s_memtime = shader cycles; sums per phase of a sample of workgroups go to stats[8 + 20 * wave ..] (vlq_ivfpq_stats prints
// them); no result is computed from a stamp.  The build also waits for the prefetched row explicitly before the table build,
// so that waiting for memory and storing the table are two phases.
// Ablation builds (tools/build_variant.sh ablN "-DVLQ_SCAN16_ABL=N" scan16): kernel TIME with one part of the probe loop removed --
// results are wrong under every one of them.  Bits: 1 no selection, 2 no gathers, 4 no table stores, 8 no row loads, 16 no code reloads
#ifndef VLQ_SCAN16_ABL
#define VLQ_SCAN16_ABL 0
#endif
static constexpr bool kAblSelect = (VLQ_SCAN16_ABL & 1) != 0, kAblGather = (VLQ_SCAN16_ABL & 2) != 0, kAblStore = (VLQ_SCAN16_ABL & 4) != 0,
                      kAblRows = (VLQ_SCAN16_ABL & 8) != 0, kAblCodes = (VLQ_SCAN16_ABL & 16) != 0;
#ifdef VLQ_SCAN16_PHASES
#define VLQ_PH_DECL                                              \
    uint64_t ph_[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    uint64_t ph_last_ = __builtin_amdgcn_s_memtime();            \
    const uint64_t ph_t0_ = ph_last_, ph_rt0_ = wall_clock64();  \
    uint32_t ph_trips_ = 0;                                      \
    uint64_t ph_g_ = 0, ph_g0_ = 0
#define VLQ_PH(i)                                                \
    do {                                                         \
        asm volatile("" ::: "memory");                           \
        const uint64_t n_ = __builtin_amdgcn_s_memtime();        \
        asm volatile("" ::: "memory");                           \
        ph_[i] += n_ - ph_last_;                                 \
        ph_last_ = n_;                                           \
    } while (0)
#define VLQ_PH_VMWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define VLQ_PH_TRIP() ph_trips_++
#define VLQ_PH_G0() do { asm volatile("" ::: "memory"); ph_g0_ = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#define VLQ_PH_G1() do { asm volatile("" ::: "memory"); ph_g_ += __builtin_amdgcn_s_memtime() - ph_g0_; asm volatile("" ::: "memory"); } while (0)
#else
#define VLQ_PH_G0() do {} while (0)
#define VLQ_PH_G1() do {} while (0)
#define VLQ_PH_DECL
#define VLQ_PH(i) do {} while (0)
#define VLQ_PH_VMWAIT() do {} while (0)
#define VLQ_PH_TRIP() do {} while (0)
#endif

// IMI: table type 2 (multi-index: two term2 rows per list) -- a compile-time switch, the row
// addressing sits in the per-probe prefetch
// OWNED: the list-owned schedule (kernels.h): the workgroup is one (query, list partition) item
template <int KPL, int NW, int NBUF, bool PIPE, bool IMI, bool OWNED = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu((((KPL == 4 || KPL == 2) && PIPE) || (NW == 2 && KPL <= 2)) ? 4 : 1))) void scan16_kernel(ScanArgs a, int lut_region) {
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
    // nprobe <= 64 (round 5): the probes' metadata once more, in WALKING order -- list offsets and scan positions here, lengths /
    // list ids / coarse distances permuted in place in pm -- so that the probe loop reads probe i's five values at index i
    // in one LDS round trip instead of ord[i] -> p -> pm.*[p] in two (the round trips wait behind the CU's gathers)
    const bool recs = a.nprobe <= 64;
    int64_t* w_off = reinterpret_cast<int64_t*>((reinterpret_cast<uintptr_t>(wg_thr + 1) + 7) & ~(uintptr_t)7);   // [nprobe]
    uint32_t* w_pos = reinterpret_cast<uint32_t*>(w_off + a.nprobe);                                              // [nprobe]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    VLQ_PH_DECL;
    // adc16_fixed() addresses the LUT buffers at LDS offsets 0 / 16384
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }
    uint32_t two = 2;
    asm volatile("" : "+v"(two));   // keep the shift amount in a VGPR (SDWA takes no literal)
    // XCD-aware placement: hardware deals consecutive workgroups round-robin over the 8
    // XCDs, so give XCD x the x-th contiguous chunk of the (sorted) query order.
    // small batches: a query's probes are split over a.nsplit workgroups (parts = contiguous ranges
    // of the walking order) that write partial top-k rows [part][nq][k]; merge_topk_kernel joins them
    int64_t q;
    int part = 0, own_x = 0, nparts = 1, tail_slot = -1;
    if (OWNED) {
        // consecutive workgroups go round-robin over the 8 XCDs: XCD x serves partition x, its items in
        // the order launch_owned_order gave them (neighbouring lists next to each other in time)
        const int64_t b = blockIdx.x;
        own_x = (int)(b & 7);
        const int64_t slot = b >> 3;
        if (slot >= a.own_count[own_x]) return;
        q = a.own_order[(int64_t)own_x * a.nq + slot];
    } else if (a.tail_r > 0) {
        // split tail (kernels.h): XCD x owns the sorted positions [x * chunk, (x + 1) * chunk) /\ [0, nq); the first of them
        // whole, one workgroup each, the last tail_r in tail_p parts -- dispatched after the whole ones, they fill the
        // slots of the batch's last, partial round with short workgroups instead of leaving most of the chip idle
        const int64_t b = blockIdx.x;
        const int x = (int)(b & 7);
        const int64_t l = b >> 3;
        const int64_t first = (int64_t)x * a.xcd_chunk;
        const int64_t nx = max((int64_t)0, min((int64_t)a.xcd_chunk, a.nq - first));
        const int64_t tx = min((int64_t)a.tail_r, nx), wx = nx - tx;
        int64_t qs;
        if (l < wx) {
            qs = first + l;
        } else {
            const int64_t l2 = l - wx;
            if (l2 >= tx * a.tail_p) return;
            const int64_t ti = l2 / a.tail_p;
            part = (int)(l2 - ti * a.tail_p);
            nparts = a.tail_p;
            qs = first + wx + ti;
            tail_slot = (int)((int64_t)x * a.tail_r + ti);
        }
        q = a.qorder ? a.qorder[qs] : qs;
        if (tail_slot >= 0 && part == 0 && threadIdx.x == 0) a.tail_rows[tail_slot] = (int)q;
    } else {
        const int64_t b = blockIdx.x;
        const int64_t s = (b & 7) * a.xcd_chunk + (b >> 3);
        if (s >= a.nq * a.nsplit) return;
        const int64_t qs = s / a.nsplit;
        part = (int)(s - qs * a.nsplit);
        nparts = a.nsplit;
        q = a.qorder ? a.qorder[qs] : qs;
    }
    const int64_t* kq = a.keys + q * a.nprobe;

    // ---- per-query set-up -------------------------------------------------------
    VLQ_PH(8);
    const bool badkey = probe_meta_fill(a, q, pm, t, NT);
    VLQ_PH(9);
    float4 m2t3[NI];
    load_query_table16<NI>(a, q, t, lane, wave, m2t3);
    WalkPre wpre;
    if (!OWNED && wave == 0) wpre = walk_prefetch(a, lane);       // in flight across the barrier and the prefix sums
    VLQ_PH(10);
    __syncthreads();
    VLQ_PH(11);
    int walk_mean = -1;        // thread 0: walk_order.cuh
    if (wave == 0) {
        const int cut = probe_meta_scan(a, pm, lane);
        __builtin_amdgcn_wave_barrier();
        int nl = 0;
        for (int p0 = 0; p0 < cut; p0 += 64) {      // coarse-distance order, dead probes dropped
            const int p = p0 + lane;
            bool lv = p < cut && pm.pkey[p] >= 0;
            if (OWNED) lv = lv && a.list_part[pm.pkey[p]] == own_x;      // this item's share of the probes
            const u64 mask = __ballot(lv);
            if (lv) ord[nl + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)p;
            nl += __popcll(mask);
        }
        VLQ_PH(12);
        if (!OWNED && nparts == 1) walk_mean = walk_order_sort(a, pm, ord, nl, lane, wpre);    // parts are merged in part order = scan order
        if (recs) {
            const int p = lane < nl ? ord[lane] : 0;
            const int32_t k_ = pm.pkey[p];
            const uint32_t l_ = pm.plen[p], c_ = pm.cum[p];
            const float d_ = pm.pd0[p];
            const int64_t o_ = pm.poff[p];
            __builtin_amdgcn_wave_barrier();           // (LDS operations of one wave complete in order: every read above precedes the writes)
            if (lane < nl) { pm.pkey[lane] = k_; pm.plen[lane] = l_; pm.pd0[lane] = d_; w_off[lane] = o_; w_pos[lane] = c_; }
        }
        if (lane == 0) { misc[0] = cut; misc[1] = nl; *wg_thr = f32_to_ordered(3.402823466e+38f); }
    }
    VLQ_PH(13);
    __syncthreads();
    const int nlive = misc[1];

    WaveSelect<KPL, QR, KPL >= 2> sel;   // k > 64: the merge network stays out of the scan loop's register budget
    sel.init(a.k, queue + wave * 64 * QR, lane);

    // ---- probe loop, software-pipelined one live probe ahead ----------------------
    float4 t2r[NI];
    // The codes of a list are requested one PROBE ahead, chunk by chunk (round 5): cr[c] holds chunk c (this thread's code of
    // trip c) of the list about to be scanned and is reloaded with the next list's chunk c in the trip that consumed it.  Until
    // round 4 only the first chunk came a probe ahead and every later one a single trip ahead: a trip's 16 gathers and adds
    // take ~580 cycles, a code load under the scan's fabric traffic ~1500 -- every trip but the first of a list waited ~900
    // cycles for its codes (profiles/r05_scan16_phases.txt).  Same register count: three chunks in flight either way.
    // (the long selections, KPL >= 4, keep one chunk a probe ahead and the rest a trip ahead: their merge networks leave no
    // registers for more)
    constexpr bool AHEAD = !PIPE && KPL <= 2;
    constexpr int NPRE = !AHEAD ? 2 : (NW == 2 ? 3 : 2);
    uint4 cr[3] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    // the prefetched probe's metadata is carried into the iteration that scans it (wave-uniform
    // values): the loop top has no LDS round trips of its own
    uint32_t n_len = 0, n_pos0 = 0;
    float n_dis0 = 0.f;
    int64_t n_off = 0;
    // Two steps (round 5): the probe's metadata (wave-uniform: LDS -> scalar registers) while the table is being built, the
    // global loads -- the 16 KB row, the list's first codes -- only AFTER the first trip has requested the current list's
    // second chunk.  vmcnt retires in order: with the row (fabric latency, 8 loads) issued ahead of that chunk, the second
    // trip of every probe waited for the NEXT probe's row (profiles/r05_scan16_phases.txt: 1410 cycles per trip on the
    // headline's 330-code lists against 920 on G1, whose rows are L2 hits).
    int64_t n_key = 0;
    auto prefetch_meta = [&](int i) {     // i-th probe of the walking order
        if (i >= nlive) return;      // (a part may look one probe past its range: harmless loads)
        if (recs) {
            n_key = (int64_t)__builtin_amdgcn_readfirstlane(pm.pkey[i]);
            n_len = __builtin_amdgcn_readfirstlane(pm.plen[i]);
            n_dis0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pm.pd0[i])));
            n_pos0 = __builtin_amdgcn_readfirstlane(w_pos[i]);
            const int64_t o = w_off[i];
            n_off = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)o >> 32)) << 32) |
                              __builtin_amdgcn_readfirstlane((uint32_t)o));
            return;
        }
        const int p = ord[i];
        n_key = (int64_t)__builtin_amdgcn_readfirstlane(pm.pkey[p]);
        n_len = __builtin_amdgcn_readfirstlane(pm.plen[p]);
        n_dis0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pm.pd0[p])));
        n_pos0 = __builtin_amdgcn_readfirstlane(pm.cum[p]);
        {
            const int64_t o = pm.poff[p];
            n_off = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)o >> 32)) << 32) |
                              __builtin_amdgcn_readfirstlane((uint32_t)o));
        }
    };
    auto load_rows = [&]() {
        const int64_t key = n_key;
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
    };
    // this thread's code of trip c of the prefetched list, clamped (branch-free loads)
    auto load_chunk = [&](auto cc_) {
        constexpr int C = decltype(cc_)::value;
        cr[C] = (reinterpret_cast<const uint4*>(a.codes) + n_off)[min((uint32_t)t + C * NT, n_len - 1)];
    };
    auto prefetch_loads = [&](int i) {
        if (i >= nlive) return;
        load_rows();
        load_chunk(std::integral_constant<int, 0>{});
        load_chunk(std::integral_constant<int, 1>{});
        if (NPRE == 3) load_chunk(std::integral_constant<int, 2>{});
    };
    auto prefetch = [&](int i) { prefetch_meta(i); prefetch_loads(i); };
    const int i_begin = (int)((int64_t)part * nlive / nparts), i_end = (int)((int64_t)(part + 1) * nlive / nparts);
    const unsigned long long t_walk = wall_clock64();
    prefetch(i_begin);
    // (the first table build needs the row at once, so nothing is lost by waiting for the set-up's loads here -- and with
    // nothing pending on the way into the loop the compiler counts the loads younger than a row exactly: entering with the
    // prologue's loads in another order than the loop's it waited for vmcnt(0) in the middle of EVERY table build, i.e. for
    // the chunk requested in the trip just finished)
    if (AHEAD) __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    int buf = 0;
    uint64_t nscan = 0;
    VLQ_PH(0);
    for (int i = i_begin; i < i_end; i++) {
        const uint32_t len = n_len;
        const float dis0 = n_dis0;
        const uint32_t pos0 = n_pos0;
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + n_off;
        float* L = lut + buf * E;
        if (NBUF == 1) __syncthreads();   // single LUT buffer: everyone is done scanning with it
        VLQ_PH(1);
        VLQ_PH_VMWAIT();
        VLQ_PH(2);
        // the table build and the next list's first loads at raised wave priority: the workgroup's
        // other waves wait at the barrier below for the slowest builder, whose stores and global
        // loads otherwise queue behind the gathers of the CU's other workgroups (scan 0.665 ->
        // 0.655 ms at the headline shape; levels 1..3 measure the same)
        __builtin_amdgcn_s_setprio(2);
        if (kAblStore) {
#pragma unroll
            for (int i2 = 0; i2 < NI; i2++) asm volatile("" :: "v"(t2r[i2].x), "v"(t2r[i2].y), "v"(t2r[i2].z), "v"(t2r[i2].w));
        } else build_lut16<NI>(L, t, t2r, m2t3);
        uint4 cc = cr[0], cd = cr[1];
        // a list longer than NPRE chunks: its next two chunks requested now (the row registers are free once the table is
        // stored), in flight across the barrier (four until the keyed admission arrived: at 128 VGPRs the two-wave shape
        // then spilled a pair of table registers and reloaded it -- behind a vmcnt(0) -- in every table build)
        const uint32_t w64x = (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 64 + NPRE * NT;
        const bool hasx = AHEAD && w64x < len;
        constexpr int NEX = 2;
        uint4 ex[NEX];
        if (hasx) {
#pragma unroll
            for (int e = 0; e < NEX; e++) ex[e] = cp[min(w64x + e * NT + lane, len - 1)];
        }
        if (!AHEAD) prefetch(i + 1);
        else prefetch_meta(i + 1);
        __builtin_amdgcn_s_setprio(0);
        VLQ_PH(3);
        __syncthreads();
        VLQ_PH(4);
        if (sel.dirty) {     // wave-uniform: publish this wave's k-th distance, then take the workgroup's minimum
            if (lane == 0) atomicMin(wg_thr, f32_to_ordered(sel.thr_own));
            sel.dirty = false;
        }
        sel.refresh_with(*wg_thr);
        VLQ_PH(5);
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
                const bool hit_a = __builtin_amdgcn_ballot_w64(da <= sel.thr_le) != 0;
                VLQ_WAIT8(8, h3);
                float db = dis0;
#pragma unroll
                for (int m = 0; m < 8; m++) db = __fadd_rn(db, h3[m]);
                asm volatile("" : "+v"(db));
                VLQ_WAIT8(0, h4);
#pragma unroll
                for (int m = 0; m < 8; m++) db = __fadd_rn(db, h4[m]);
                if (hit_a) sel.offer_keyed(da, pos0 + ja, true);
                sel.offer_keyed(db, pos0 + jb, jb < len);
            }
            for (; !AHEAD && j0 < len; j0 += NT) {           // (the pair loop's odd last chunk)
                const uint32_t j = j0 + lane;
                const uint4 cn = cp[min(j + NT, len - 1)];
                const float dis = adc16_halves<B>(cc, dis0, two);
                sel.offer_keyed(dis, pos0 + j, j < len);
                cc = cn;
            }
            if (AHEAD) {
                // cr[C] = this thread's code of trip C, requested one probe ahead; once trip C's gathers are issued the same
                // registers are reloaded with chunk C of the NEXT list (and, with chunk 0, its row BEHIND the chunk: vmcnt
                // retires in order, the row must not sit in front of codes needed sooner).  Everything a probe loads is loaded
                // unconditionally and in one place (behind the last probe the loads re-read the current list's addresses --
                // prefetch_meta left n_off / n_key alone -- cache hits): the compiler then counts the loads younger than the
                // one it waits for exactly; with a load under a condition, two load sites for one register, or a copy of a
                // register with a load in flight it falls back to vmcnt(0) -- a wait for the chunk requested a trip ago.
                const uint32_t w64 = (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 64;
                // a list longer than NPRE chunks: its further chunks first, NEX at a time (requested after the table build; the
                // visiting order inside a list is as free as the order of the lists: walk_order.cuh)
                if (hasx) {
                    for (uint32_t jx = w64x; jx < len; jx += NEX * NT) {
                        uint4 cur[NEX];
#pragma unroll
                        for (int e = 0; e < NEX; e++) cur[e] = ex[e];
                        if (jx + NEX * NT < len) {
#pragma unroll
                            for (int e = 0; e < NEX; e++) ex[e] = cp[min(jx + (NEX + e) * NT + lane, len - 1)];
                        }
#pragma unroll
                        for (int e = 0; e < NEX; e++) {
                            const uint32_t j0e = jx + e * NT;
                            if (j0e < len) {
                                VLQ_PH_TRIP();
                                const float dis = adc16_halves<B>(cur[e], dis0, two);
                                sel.offer_keyed(dis, pos0 + j0e + lane, j0e + lane < len);
                            }
                        }
                    }
                }
                auto trip = [&](auto cc_) {
                    constexpr int C = decltype(cc_)::value;
                    const uint32_t jc = w64 + C * NT;
                    uint32_t g = jc < len ? 1u : 0u;      // (wave-uniform)
                    float lo[8], hi[8];
                    if (g && !kAblGather) {
                        VLQ_PH_TRIP();
                        VLQ_PH_G0();
                        const uint4 cur = cr[C];
                        if (B == 0) { { float (&v)[8] = lo; VLQ_G8LO_NW(0, cur.x, cur.y); } { float (&v)[8] = hi; VLQ_G8HI_NW(0, cur.z, cur.w); } }
                        else { { float (&v)[8] = lo; VLQ_G8LO_NW(16384, cur.x, cur.y); } { float (&v)[8] = hi; VLQ_G8HI_NW(16384, cur.z, cur.w); } }
                    }
                    if (!kAblCodes) load_chunk(cc_);
                    if (C == 0 && !kAblRows) load_rows();
                    g = __builtin_amdgcn_readfirstlane(g);
                    asm volatile("" : "+s"(g));        // (keeps the two halves of the trip from being threaded into two copies of the loads)
                    if (g && kAblGather) {
                        const uint4 cur = cr[C];
                        if (kAblSelect) asm volatile("" :: "v"(cur.x)); else sel.offer_keyed(dis0 + __uint_as_float(cur.x & 0x3fffffffu), pos0 + jc + lane, jc + lane < len);
                    }
                    if (g && !kAblGather) {
                        float dis = dis0;
                        VLQ_WAIT8(8, lo);
#pragma unroll
                        for (int m = 0; m < 8; m++) dis = __fadd_rn(dis, lo[m]);
                        asm volatile("" : "+v"(dis));
                        VLQ_WAIT8(0, hi);
#pragma unroll
                        for (int m = 0; m < 8; m++) dis = __fadd_rn(dis, hi[m]);
                        VLQ_PH_G1();
                        if (kAblSelect) asm volatile("" :: "v"(dis));
                        else sel.offer_keyed(dis, pos0 + jc + lane, jc + lane < len);
                    }
                };
                trip(std::integral_constant<int, 0>{});
                trip(std::integral_constant<int, 1>{});
                if (NPRE == 3) trip(std::integral_constant<int, 2>{});
            }
        };
        if (NBUF == 1 || buf == 0) scan_list(std::integral_constant<int, 0>{});
        else scan_list(std::integral_constant<int, 1>{});
        nscan += len;
        if (NBUF == 2) buf ^= 1;
        VLQ_PH(6);
    }
    if (t == 0) walk_state_finish(a, t_walk, i_end - i_begin, walk_mean);

    if (OWNED) {
        // raw keys out: scan positions are global to the query, so owned_merge_kernel can order the
        // parts' candidates exactly like one workgroup scanning all probes would have
        if (merge_waves<KPL, NW, QR>(sel, smraw, a.k, wave, lane)) {
            u64* out = a.part_keys + ((size_t)q * 8 + own_x) * a.k;
#pragma unroll
            for (int r = 0; r < KPL; r++) {
                const int e = r * 64 + lane;
                if (e < a.k) out[e] = sel.best[r];
            }
        }
    } else if (tail_slot >= 0) {
        ScanArgs em = a;             // partial rows of a split tail query: [part][8 * tail_r][k]
        em.D = a.tail_D;
        em.I = a.tail_I;
        merge_and_emit<KPL, NW, QR>(sel, smraw, pm.cum, em, (int64_t)part * 8 * a.tail_r + tail_slot, wave, lane,
                                    [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
    } else {
        merge_and_emit<KPL, NW, QR>(sel, smraw, pm.cum, a, a.nsplit > 1 ? (int64_t)part * a.nq + q : q, wave, lane,
                                    [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
    }
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
    if (badkey) *a.bad_key = 1;
#ifdef VLQ_SCAN16_PHASES
    VLQ_PH(7);
    if (!OWNED && lane == 0 && wave < 2 && (blockIdx.x % 29) == 0) {     // a sample: every workgroup's atomics would be what is measured
        unsigned long long* o = a.ncode + 8 + 20 * wave;
#pragma unroll
        for (int i2 = 0; i2 < 14; i2++) atomicAdd(o + i2, (unsigned long long)ph_[i2]);
        atomicAdd(o + 14, 1ull);
        atomicAdd(o + 15, ((unsigned long long)ph_trips_ << 32) | (unsigned long long)(i_end - i_begin));
        atomicAdd(o + 16, (unsigned long long)(ph_last_ - ph_t0_));
        atomicAdd(o + 17, (unsigned long long)(wall_clock64() - ph_rt0_));
        atomicAdd(o + 18, (unsigned long long)ph_g_);
    }
#endif
}

// joins the parts of the list-owned schedule: one wave per query rebuilds the query's probe metadata
// (the same prefix sums and max_codes cut every item used), selects the k smallest of its parts' keys
// -- a total order (distance, scan position), so the result is what one workgroup scanning all probes
// returns -- and translates positions to ids
template <int KPL>
__global__ __launch_bounds__(256) void owned_merge_kernel(ScanArgs a, int stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= a.nq) return;                      // (no workgroup barrier below)
    unsigned char* base = smraw + (size_t)wave * stride;
    u64* queue = reinterpret_cast<u64*>(base);                       // [64]
    ProbeMeta pm;
    pm.carve(base + 512, a.nprobe);
    const int64_t* kq = a.keys + q * a.nprobe;
    if (probe_meta_fill(a, q, pm, lane, 64)) *a.bad_key = 1;     // a query whose probes are all invalid has no scan workgroup
    __builtin_amdgcn_wave_barrier();
    probe_meta_scan(a, pm, lane);
    __builtin_amdgcn_wave_barrier();
    WaveSelect<KPL> sel;
    sel.init(a.k, queue, lane);
    const uint32_t m = a.part_mask[q];
    for (int x = 0; x < 8; x++) {
        if (!((m >> x) & 1u)) continue;
        const u64* src = a.part_keys + ((size_t)q * 8 + x) * a.k;
        for (int e0 = 0; e0 < a.k; e0 += 64) {
            const int e = e0 + lane;
            const bool valid = e < a.k;
            sel.offer_key(valid ? src[e] : kMaxKey, valid);
        }
    }
    sel.flush();
    emit_rows<KPL>(sel, pm.cum, a, q, lane, [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
}

void launch_owned_merge(const ScanArgs& a, hipStream_t s) {
    if (a.nq <= 0) return;
    const int stride = (int)((512 + (size_t)a.nprobe * 24 + 8 + 15) & ~(size_t)15);
    const size_t smem = (size_t)4 * stride;
    dim3 grid((unsigned)((a.nq + 3) / 4)), block(256);
#define VLQ_OM(K)                                                                                  \
    do {                                                                                           \
        ensure_dynamic_lds(reinterpret_cast<const void*>(owned_merge_kernel<K>), smem);            \
        hipLaunchKernelGGL(owned_merge_kernel<K>, grid, block, smem, s, a, stride);                \
    } while (0)
    if (a.k <= 64) VLQ_OM(1);
    else if (a.k <= 128) VLQ_OM(2);
    else if (a.k <= 256) VLQ_OM(4);
    else if (a.k <= 512) VLQ_OM(8);
    else VLQ_OM(16);
#undef VLQ_OM
}

// ---------------------------------------------------------------------------
// list-owned schedule, preparation: for every query the partitions its probes fall into, and per
// partition the queries in the order of the spatial rank of their NEAREST list of that partition.
// Ordering only decides which items run next to each other; never a result.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void owned_hist_kernel(const int64_t* __restrict__ keys, int64_t nq, int nprobe,
                                                         int nlist, const int* __restrict__ list_rank,
                                                         const uint8_t* __restrict__ list_part, int* __restrict__ hist,
                                                         int* __restrict__ minr, uint8_t* __restrict__ part_mask) {
    __shared__ int mr[256][9];                 // (row stride 9: bank-conflict free)
    const int t = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * 256 + t;
    if (q >= nq) return;
#pragma unroll
    for (int x = 0; x < 8; x++) mr[t][x] = 0x7fffffff;
    for (int p = 0; p < nprobe; p++) {
        const int64_t key = keys[q * nprobe + p];
        if (key < 0 || key >= nlist) continue;
        const int x = list_part[key], r = list_rank[key];
        if (r < mr[t][x]) mr[t][x] = r;
    }
    uint32_t m = 0;
#pragma unroll
    for (int x = 0; x < 8; x++) {
        const int r = mr[t][x];
        minr[q * 8 + x] = r;
        if (r != 0x7fffffff) { m |= 1u << x; atomicAdd(&hist[(size_t)x * nlist + r], 1); }
    }
    part_mask[q] = (uint8_t)m;
}

__global__ __launch_bounds__(256) void owned_place_kernel(int64_t nq, int nlist, const int* __restrict__ hist,
                                                          int* __restrict__ cnt, const int* __restrict__ minr,
                                                          int* __restrict__ own_order, int* __restrict__ own_count) {
    extern __shared__ int pre[];                 // [nlist] exclusive prefix of this partition's bins
    __shared__ int part[256];
    const int t = threadIdx.x, x = blockIdx.y;
    const int* hx = hist + (size_t)x * nlist;
    const int per = (nlist + 255) / 256;
    const int b0 = t * per;
    int sum = 0;
    for (int i = 0; i < per; i++) if (b0 + i < nlist) sum += hx[b0 + i];
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
        if (b0 + i < nlist) { pre[b0 + i] = run; run += hx[b0 + i]; }
    if (blockIdx.x == 0 && t == 255) own_count[x] = part[255];
    __syncthreads();
    const int64_t q = (int64_t)blockIdx.x * 256 + t;
    if (q >= nq) return;
    const int r = minr[q * 8 + x];
    if (r == 0x7fffffff) return;
    own_order[(int64_t)x * nq + pre[r] + atomicAdd(&cnt[(size_t)x * nlist + r], 1)] = (int)q;
}

void launch_owned_order(const int64_t* keys, int64_t nq, int nprobe, int nlist, const int* list_rank,
                        const uint8_t* list_part, int* hist, int* minr, int* own_order, int* own_count,
                        uint8_t* part_mask, hipStream_t s) {
    if (nq <= 0) return;
    (void)hipMemsetAsync(hist, 0, (size_t)16 * nlist * sizeof(int), s);     // hist | cnt
    const unsigned g = (unsigned)((nq + 255) / 256);
    hipLaunchKernelGGL(owned_hist_kernel, dim3(g), dim3(256), 0, s, keys, nq, nprobe, nlist, list_rank, list_part,
                       hist, minr, part_mask);
    const size_t smem = (size_t)nlist * sizeof(int);
    ensure_dynamic_lds(reinterpret_cast<const void*>(owned_place_kernel), smem);
    hipLaunchKernelGGL(owned_place_kernel, dim3(g, 8), dim3(256), smem, s, nq, nlist, hist, hist + (size_t)8 * nlist,
                       minr, own_order, own_count);
}

// per-query table of the list-owned schedule: out[q][m][j] = (-2) * <q_m, cent_mj> with exactly the
// operations of load_query_table16's fused form (fvec_inner_product order, utils.cpp:509-533).  A
// workgroup keeps one quarter of the transposed codebook in registers and runs over QB queries.
__global__ __launch_bounds__(256) void qtab16_kernel(const float* __restrict__ queries, int64_t nq,
                                                     const float* __restrict__ pq_cent_t, float* __restrict__ qtab) {
    constexpr int QB = 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i = blockIdx.y;                   // quarter: sub-quantizers 4*i .. 4*i+3, one per wave
    const int m = 4 * i + wave;
    const float4* ct = reinterpret_cast<const float4*>(pq_cent_t + (size_t)m * 8 * 256) + lane;
    const float4 y0 = ct[0 * 64], y1 = ct[1 * 64], y2 = ct[2 * 64], y3 = ct[3 * 64];
    const float4 y4 = ct[4 * 64], y5 = ct[5 * 64], y6 = ct[6 * 64], y7 = ct[7 * 64];
    const int64_t q0 = (int64_t)blockIdx.x * QB;
    for (int qi = 0; qi < QB; qi++) {
        const int64_t q = q0 + qi;
        if (q >= nq) break;
        const float* qv = queries + q * 128;
        const float4 x0 = *reinterpret_cast<const float4*>(qv + m * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(qv + m * 8 + 4);
#define VLQ_IP8(C)                                                                                        \
    __fmul_rn(-2.f,                                                                                      \
              __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.x, y0.C)), __fmul_rn(x1.x, y4.C)), 0.f), \
                                  __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.y, y1.C)), __fmul_rn(x1.y, y5.C)), 0.f)), \
                        __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.z, y2.C)), __fmul_rn(x1.z, y6.C)), 0.f), \
                                  __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.w, y3.C)), __fmul_rn(x1.w, y7.C)), 0.f))))
        reinterpret_cast<float4*>(qtab + q * 4096)[i * 256 + t] = make_float4(VLQ_IP8(x), VLQ_IP8(y), VLQ_IP8(z), VLQ_IP8(w));
#undef VLQ_IP8
    }
}

void launch_qtab16(const float* queries, int64_t nq, const float* pq_cent_t, float* qtab, hipStream_t s) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(qtab16_kernel, dim3((unsigned)((nq + 15) / 16), 4), dim3(256), 0, s, queries, nq, pq_cent_t, qtab);
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
        // Multi-index cells share their halves: a query's 64 cells come from ~20 x ~20 sub-centroids, and each half's 8 KB of
        // term2 is what crosses the fabric (profiles/r06_scan16_short_pmc.txt).  The cells are walked in (first half, second
        // half) order so that the four waves work on cells of one first half at a time and a second half comes back a few
        // cells later, not a whole walk later.  Scan positions are fixed per probe (pm.cum), so the results do not move.
        static_assert(sizeof(u64) == 8, "");
        if (a.imi_nbits > 0 && nl <= 64 && nl > 1 && !a.short_keep_order) {
            __builtin_amdgcn_wave_barrier();
            const int p = lane < nl ? ord[lane] : 0;
            const int64_t key = lane < nl ? (int64_t)pm.pkey[p] : 0;
            const u64 i0 = (u64)(key & ((int64_t(1) << a.imi_nbits) - 1)), i1 = (u64)(key >> a.imi_nbits);
            const u64 sk = lane < nl ? ((i0 << 40) | (i1 << 16) | (u64)p) : kMaxKey;
            const u64 so = wave_sort64(sk, lane);
            __builtin_amdgcn_wave_barrier();
            if (lane < nl) ord[lane] = (uint16_t)(so & 0xffffu);
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
            sel.offer_keyed(dis, pos0 + j, j < len);
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
    static const bool keep_order = getenv("VLQ_SHORT_KEEP_ORDER") != nullptr;
    a.short_keep_order = keep_order ? 1 : 0;
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
    const unsigned grid = (unsigned)(8 * a.grid_per_xcd);
    hipLaunchKernelGGL((scan16_kernel<KPL, NW, NBUF, PIPE, IMI>), dim3(grid), dim3(64 * NW), smem, s, a, lut_region);
}
// the shape of the last scan16 launch of this thread (vlq_ivfpq_last_scan_info: tests pin the default path with it)
static thread_local char g_last_scan16[64] = "";
const char* last_scan16_shape() { return g_last_scan16; }

template <int KPL, int NW, int NBUF, bool PIPE>
static void launch_scan16_t(const ScanArgs& a, int lut_region, size_t smem, hipStream_t s) {
    snprintf(g_last_scan16, sizeof(g_last_scan16), "scan16_kernel<%d, %d, %d, %s, %s, %s>", KPL, NW, NBUF, PIPE ? "true" : "false",
             a.imi_nbits > 0 ? "true" : "false", a.part_keys ? "true" : "false");
    if (a.part_keys) {        // list-owned schedule: 8 x nq slots, the surplus exits at once
        ensure_dynamic_lds(reinterpret_cast<const void*>(scan16_kernel<KPL, NW, NBUF, PIPE, false, true>), smem);
        hipLaunchKernelGGL((scan16_kernel<KPL, NW, NBUF, PIPE, false, true>), dim3((unsigned)(8 * a.nq)), dim3(64 * NW), smem, s,
                           a, lut_region);
        return;
    }
    if (a.imi_nbits > 0) launch_scan16_i<KPL, NW, NBUF, PIPE, true>(a, lut_region, smem, s);
    else launch_scan16_i<KPL, NW, NBUF, PIPE, false>(a, lut_region, smem, s);
}

void launch_scan16_owned(const ScanArgs& a, hipStream_t s) { launch_scan16(a, s); }

void launch_scan16(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    if (a.nsplit < 1 || a.part_keys) a.nsplit = 1;
    a.xcd_chunk = (int)((a.nq * a.nsplit + 7) / 8);
    if (a.nsplit > 1 || a.part_keys || !a.tail_D) a.tail_r = 0;
    a.grid_per_xcd = a.xcd_chunk;
    if (a.tail_r > 0) {              // per XCD: its whole queries, then tail_p workgroups for each of its last tail_r
        if (a.tail_r > a.xcd_chunk) a.tail_r = a.xcd_chunk;
        a.grid_per_xcd = a.xcd_chunk - a.tail_r + a.tail_r * a.tail_p;
    }
    // k <= 64: 8 waves per workgroup share one LUT (32 waves per CU at 4 workgroups);
    // larger k keeps more selection state per wave, so stay at 4 waves
    // Measured alternatives (r01, MI355X, bench data): 8 waves per workgroup 0.95 ms, single
    // LUT buffer with 6 workgroups per CU 0.78 ms, two probes of lookahead 0.82 ms, two
    // queries per workgroup sharing term2 rows 0.93 ms -- against 0.78-0.82 ms for this
    // configuration (4 waves, double-buffered LUT, one probe of lookahead).
    // (r03, against 0.649 ms: single LUT buffer without the pair loop = 99 VGPRs = 5 workgroups per CU 0.682 ms, single
    // buffer with the pair loop at 4 per CU 0.668 ms.)
    const int nw = 4;
    size_t lutb = (size_t)2 * 4096 * 4;
    const size_t merge = (size_t)nw * a.k * 8;
    if (lutb < merge) lutb = merge;
    const size_t wrec = a.nprobe <= 64 ? (size_t)a.nprobe * 12 + 8 : 0;       // walking-order copies of the probe metadata (scan16_kernel)
    const size_t tail = (size_t)nw * 64 * 8 * (a.k > 256 ? 4 : 1) + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 2 + 8 + 64 + wrec;
    const size_t smem = lutb + tail;
    // k <= 64.  Lists of a few hundred codes (mean list < 1024 codes: every BASELINE shape but the long-list tools): ONE table
    // buffer and the plain chunk loop -- 95 VGPRs and 19 KB of LDS = 5 workgroups per CU instead of 4.  Round 4, 10 000
    // queries: G1 data 0.649 -> 0.606 ms (0.97 of the LDS gather rate), headline data 0.730 = 0.730 (row traffic bound); 2500
    // queries 0.230 -> 0.215 / 0.192 -> 0.179 ms: 1280 slots hold a sharded batch's slice in fewer rounds.  Long lists keep two
    // buffers and the pair loop (two chunks per trip, the adds of one under the gathers of the other).
    // VLQ_SCAN16_VARIANT (A/B): 0 = two buffers + pair loop always, 2 = two buffers, plain loop, 3 = one buffer, pair loop.
    static const int variant = [] { const char* e = getenv("VLQ_SCAN16_VARIANT"); return e ? atoi(e) : -1; }();
    const bool plain = !a.part_keys && a.imi_nbits == 0;
    // ... and from 3000 queries on TWO waves per workgroup (a thread owns 32 table entries: 127 VGPRs, 8 workgroups per CU = 2048
    // slots): a list of 330 codes is 3 trips of 128 lanes instead of 2 trips of 256 -- 25 % fewer lane slots, and a two-wave
    // barrier.  10 000 queries: headline data 0.604 -> 0.548 ms, nprobe 16 / 64 / 128 0.356 / 1.08 / 2.04 -> 0.327 / 0.99 / 1.92,
    // k = 50 0.649 -> 0.563, G1 0.626 -> 0.595; below 3000 queries the 2048 slots of slower workgroups lose to 1280 (2500
    // queries 0.183 -> 0.202, 1250 queries on G1 0.101 -> 0.116).  VLQ_SCAN16_VARIANT = 4 / 1 force two / four waves.
    if (a.k <= 64 && plain && !a.tail_r && a.nsplit == 1 && (variant == 4 || (variant < 0 && !a.long_lists && a.nq >= 3000))) {
        const size_t l1 = std::max((size_t)4096 * 4, (size_t)2 * a.k * 8);
        const size_t tail2 = (size_t)2 * 64 * 8 + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 2 + 8 + 64 + wrec;
        launch_scan16_t<1, 2, 1, false>(a, (int)l1, l1 + tail2, s);
    } else if (a.k <= 64 && plain && (variant == 1 || (variant < 0 && !a.long_lists))) {
        const size_t l1 = std::max((size_t)4096 * 4, merge);
        launch_scan16_t<1, 4, 1, false>(a, (int)l1, l1 + tail, s);
    } else if (a.k <= 64 && variant == 6 && plain && !a.tail_r && a.nsplit == 1) {
        // two waves AND two table buffers (one barrier per probe; 33 KB of LDS = 4 workgroups per CU)
        const size_t l2 = std::max((size_t)2 * 4096 * 4, (size_t)2 * a.k * 8);
        const size_t tail2 = (size_t)2 * 64 * 8 + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 2 + 8 + 64 + wrec;
        launch_scan16_t<1, 2, 2, false>(a, (int)l2, l2 + tail2, s);
    } else if (a.k <= 64 && variant == 2 && plain) launch_scan16_t<1, 4, 2, false>(a, (int)lutb, smem, s);
    else if (a.k <= 64 && variant == 3 && plain) {
        const size_t l1 = std::max((size_t)4096 * 4, merge);
        launch_scan16_t<1, 4, 1, true>(a, (int)l1, l1 + tail, s);
    } else if (a.k <= 64) launch_scan16_t<1, 4, 2, true>(a, (int)lutb, smem, s);
    else if (a.k <= 128 && plain && !a.tail_r && a.nsplit == 1 && (variant == 5 || (variant < 0 && !a.long_lists && a.nq >= 3000))) {
        // two waves per workgroup as for k <= 64 (128 VGPRs forced): k = 100, 10 000 queries: headline data 0.815 -> 0.77 ms,
        // nprobe 64 1.34 -> 1.13, G1 0.75 -> 0.68
        const size_t l1 = std::max((size_t)4096 * 4, (size_t)2 * a.k * 8);
        const size_t tail2 = (size_t)2 * 64 * 8 + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 2 + 8 + 64 + wrec;
        launch_scan16_t<2, 2, 1, false>(a, (int)l1, l1 + tail2, s);
    }
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
                                                           const int* __restrict__ list_rank, int shift, int nbins, int* walk_freeze) {
    extern __shared__ int pre[];                 // [nbins] exclusive prefix
    // a search that does not re-sample the walking statistic still freezes this launch's clock period per XCD = the running
    // mean of the walk times measured so far (walk_stat_kernel does it otherwise)
    if (walk_freeze && blockIdx.x == 0 && threadIdx.x < 8) {
        const int mean = walk_freeze[threadIdx.x * 16];
        if (mean > 0) walk_freeze[threadIdx.x * 16 + 1] = mean;
    }
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

// One workgroup does the whole counting sort in LDS when the batch is small (<= 2048 queries: the slices of a
// batch sharded over several GPUs): bin counts, their exclusive prefix and the placement -- one launch
// instead of a memset and two kernels: 15 -> 8 us at 1250 queries.  (At 10 000 queries one workgroup is too
// serial: 33 us against 17.)
// one (pair of neighbours in the scan order, probe of the first) sample of walk_stat_kernel (below)
__device__ __forceinline__ int walk_stat_sample(const int64_t* __restrict__ keys, const int* qorder, int nq, int nprobe, int pairs,
                                                int pair, int i) {
    const int np = min(nprobe, 32);
    if (pair >= pairs || i >= np) return 0;
    const int s0 = (int)((int64_t)pair * (nq - 1) / pairs);
    const int64_t x = keys[(int64_t)qorder[s0] * nprobe + i];
    const int64_t* kb = keys + (int64_t)qorder[s0 + 1] * nprobe;
    bool hit = false;
#pragma unroll 8
    for (int j = 0; j < np; j++) hit = hit || (kb[j] == x);
    return (hit && x >= 0) ? 1 : 0;
}

__global__ __launch_bounds__(1024) void qorder_single_kernel(const int64_t* __restrict__ keys, int nq, int nprobe, int nlist,
                                                             int* __restrict__ qorder, const int* __restrict__ list_rank,
                                                             int shift, int nbins, int* walk_freeze) {
    extern __shared__ int cnt[];                 // [nbins] counts, then running offsets
    if (walk_freeze && threadIdx.x < 8) {        // (see qorder_place_kernel)
        const int mean = walk_freeze[threadIdx.x * 16];
        if (mean > 0) walk_freeze[threadIdx.x * 16 + 1] = mean;
    }
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int b = t; b < nbins; b += 1024) cnt[b] = 0;
    __syncthreads();
    auto bin_of = [&](int q) {
        const int64_t k0 = keys[(int64_t)q * nprobe];
        const bool ok = k0 >= 0 && k0 < nlist;
        return ok ? ((list_rank ? list_rank[k0] : (int)k0) >> shift) : nbins - 1;
    };
    for (int q = t; q < nq; q += 1024) atomicAdd(&cnt[bin_of(q)], 1);
    __syncthreads();
    // exclusive prefix: a thread owns `per` consecutive bins
    const int per = (nbins + 1023) / 1024;
    const int b0 = t * per;
    int sum = 0;
    for (int i = 0; i < per; i++) if (b0 + i < nbins) sum += cnt[b0 + i];
    int incl = sum;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const int o = __shfl_up(incl, sft, 64);
        if (lane >= sft) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; w++) wbase += wsum[w];
    int run = wbase + incl - sum;
    for (int i = 0; i < per; i++)
        if (b0 + i < nbins) { const int c = cnt[b0 + i]; cnt[b0 + i] = run; run += c; }
    __syncthreads();
    for (int q = t; q < nq; q += 1024) qorder[atomicAdd(&cnt[bin_of(q)], 1)] = q;
}

// at most 16 Ki bins (the prefix is recomputed per workgroup in LDS): many-list indexes are binned
// by the high bits of the list id / rank -- for a multi-index key that is its second sub-index
void query_order_bins(int nlist, int* shift_out, int* nbins_out) {
    int shift = 0;
    while (((int64_t)nlist >> shift) > 16384) shift++;
    *shift_out = shift;
    *nbins_out = (int)(((int64_t)nlist - 1) >> shift) + 2;        // last bin: invalid keys
}

void launch_query_order(const int64_t* keys, int64_t nq, int nprobe, int nlist, int* hist,
                        int* qorder, hipStream_t s, const int* list_rank, int* walk_part, int* walk_state, WalkSeed seed, bool run_walk_stat,
                        bool hist_ready) {
    if (nq <= 0) return;
    int shift, nbins;
    query_order_bins(nlist, &shift, &nbins);
    if (nq <= 2048) {
        const size_t smem1 = (size_t)nbins * sizeof(int);
        ensure_dynamic_lds(reinterpret_cast<const void*>(qorder_single_kernel), smem1);
        hipLaunchKernelGGL(qorder_single_kernel, dim3(1), dim3(1024), smem1, s, keys, (int)nq, nprobe, nlist, qorder, list_rank,
                           shift, nbins, (walk_part && !run_walk_stat) ? walk_state : nullptr);
        // (the statistic inside this one-workgroup kernel was measured: 8192 samples on one CU cost 36 us against 5)
        if (walk_part && nq >= 2 && run_walk_stat) launch_walk_stat(keys, qorder, nq, nprobe, walk_part, walk_state, s, seed);
        return;
    }
    const size_t stride = query_order_bins_padded(nlist);               // hist | cnt, one aligned memset
    const unsigned g = (unsigned)((nq + 255) / 256);
    if (!hist_ready) {          // (otherwise the coarse stage's last kernel left the counts: OrderHist, kernels.h)
        (void)hipMemsetAsync(hist, 0, 2 * stride * sizeof(int), s);
        hipLaunchKernelGGL(qorder_hist_kernel, dim3(g), dim3(256), 0, s, keys, nq, nprobe, nlist, hist, list_rank,
                           shift, nbins);
    }
    const size_t smem = (size_t)nbins * sizeof(int);
    ensure_dynamic_lds(reinterpret_cast<const void*>(qorder_place_kernel), smem);
    hipLaunchKernelGGL(qorder_place_kernel, dim3(g), dim3(256), smem, s, keys, nq, nprobe, nlist, hist,
                       hist + stride, qorder, list_rank, shift, nbins, (walk_part && !run_walk_stat) ? walk_state : nullptr);
    if (walk_part && run_walk_stat) launch_walk_stat(keys, qorder, nq, nprobe, walk_part, walk_state, s, seed);
}

// ---------------------------------------------------------------------------
// Which walking order (walk_order.cuh)?  Neighbours of the scan order that probe mostly the SAME lists (dense clusters: the
// G1 generator setting, 750 shared probes of 1000) find each other's table rows in L2 anyway and lose the early admission
// bound in list-id order (+2..5 % measured); neighbours that share few lists (150 of 1000 on the headline data) are the case
// the list-id order is for (-12 %).  256 neighbour pairs of the scan order are sampled, one thread per (pair, probe of the
// first query) among each one's nearest min(nprobe, 32); a workgroup leaves the count of its 8 pairs in part[blockIdx.x],
// and the scan kernels add the 32 counts up themselves (walk_order_sort): no atomics, no zeroing, no host round trip.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void walk_stat_kernel(const int64_t* __restrict__ keys, const int* __restrict__ qorder, int nq,
                                                        int nprobe, int pairs, int* __restrict__ part, int* __restrict__ walk_state,
                                                        WalkSeed seed) {
    __shared__ int red[4];
    __shared__ unsigned long long lsum[4];
    __shared__ int lcnt[4];
    const int t = threadIdx.x;
    // this launch's clock period per XCD = the running mean of the walk times measured so far (walk_order.cuh) -- or, before
    // anything was measured (the first search of a handle: what a driver that calls search once gets), a MODEL of it: a walk is
    // nprobe probes, each costing the chip c0 + c1 x (codes of the list) while `slots` workgroups share it:
    //     period [10 ns ticks] = 0.75 x nprobe x slots x (1.36 ns + 0.0008 ns x mean list length) / 10
    // (fitted to the two bench data sets: 1.63 ns per probe at 330 codes, 1.92 at 700; a period 20-50 % off costs 1-4 % of
    // what the right one gains, no period at all costs 20 %: profiles/r05_cold_path.txt).  The lengths are those of block 0's
    // 256 sampled probes.
    if (walk_state && blockIdx.x == 0) {
        unsigned long long len = 0;
        int ok = 0;
        if (seed.slots > 0 && seed.list_off) {
            const int np = min(nprobe, 32), pair = t >> 5, i = t & 31;
            if (pair < pairs && i < np) {
                const int s0 = (int)((int64_t)pair * (nq - 1) / pairs);
                const int64_t x = keys[(int64_t)qorder[s0] * nprobe + i];
                if (x >= 0 && x < seed.nlist) { len = (unsigned long long)(seed.list_len ? seed.list_len[x] : seed.list_off[x + 1] - seed.list_off[x]); ok = 1; }
            }
        }
        for (int off = 32; off > 0; off >>= 1) { len += __shfl_down(len, off); ok += __shfl_down(ok, off); }
        if ((t & 63) == 0) { lsum[t >> 6] = len; lcnt[t >> 6] = ok; }
        __syncthreads();
        if (t < 8) {
            const int mean = walk_state[t * 16];
            int period = mean;
            const int n_ok = lcnt[0] + lcnt[1] + lcnt[2] + lcnt[3];
            if (mean <= 0 && seed.slots > 0 && n_ok > 0) {
                const float mean_len = (float)(lsum[0] + lsum[1] + lsum[2] + lsum[3]) / (float)n_ok;
                period = (int)(0.075f * (float)nprobe * (float)seed.slots * (1.36f + 0.0008f * mean_len));
            }
            walk_state[t * 16 + 1] = period;
        }
    }
    int shared = walk_stat_sample(keys, qorder, nq, nprobe, pairs, blockIdx.x * 8 + (t >> 5), t & 31);
    for (int off = 32; off > 0; off >>= 1) shared += __shfl_down(shared, off);
    if ((t & 63) == 0) red[t >> 6] = shared;
    __syncthreads();
    if (t == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

void preload_scan16_kernels() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(qorder_hist_kernel));
}

// returns the number of (pair, probe) samples behind the 32 counts in part[]
int walk_stat_samples(int64_t nq, int nprobe) { return (int)std::min<int64_t>(256, nq - 1) * std::min(nprobe, 32); }

int launch_walk_stat(const int64_t* keys, const int* qorder, int64_t nq, int nprobe, int* part, int* walk_state, hipStream_t s, WalkSeed seed) {
    const int pairs = (int)std::min<int64_t>(256, nq - 1);
    hipLaunchKernelGGL(walk_stat_kernel, dim3(32), dim3(256), 0, s, keys, qorder, (int)nq, nprobe, pairs, part, walk_state, seed);
    return pairs * std::min(nprobe, 32);
}

}  // namespace vlq
