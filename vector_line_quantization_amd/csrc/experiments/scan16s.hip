// "Stream" form of scan16_kernel (M = 16, ksub = 256, table mode 1, k <= 64): the same arithmetic and
// the same list loop, but a PERSISTENT workgroup that walks query after query and overlaps the fixed
// costs around a query's probes with the probes themselves:
//   * a workgroup claims its next query from a per-XCD counter late in the current one (six probes
//     before its end: a claimed query never waits long, so the tail stays balanced);
//   * the next query's probe metadata (keys -> list offsets / lengths -> prefix sums, walking order)
//     is staged into a second LDS slot during the current query's last probes, its first table row and
//     codes are requested with the "one probe ahead" prefetch of the current query's last probe, and its
//     -2<q, cent> part is computed into the registers the last table build has just released;
//   * the finished query's four per-wave lists are parked in LDS and merged + emitted by ONE wave
//     (rotating) after the next barrier, while the other three waves already scan the next query.
// scan16_kernel pays about 10 us of such latency chains per 68 us query.
// Results cannot differ from scan16_kernel: same tables, same sums, the same total order of keys.
//
// STATUS (round 2): EXPERIMENT, built only into libvlq_exp.so (VLQ_SCAN16=s / s0).  Bit-identical results
// and ncode on the first run, no hang -- but 1.23 ms against 0.70 ms for scan16_kernel on the bench data,
// overlapped or not: at the 128 registers that 4 workgroups per CU allow, the extra control state of the
// persistent loop costs 261 SGPR and 96 VGPR spills (scan16_kernel: 31 / 0), and the scratch traffic
// sits inside the probe loop.  The structure is right; it needs the register diet first.
#include <algorithm>
#include <type_traits>

#include "../kernels.h"
#include "../scan_common.cuh"
#include "../scan16_common.cuh"
#include "../wave_topk.cuh"

namespace vlq {

namespace {

// merge the four parked per-wave lists of a finished query and write its rows (one wave).  A real call:
// the selection network's registers stay out of the scan loop's allocation; only what the rows need is
// passed (no ScanArgs copy).
struct StreamOut {
    const int64_t* ids;
    const int64_t* kq;       // the query's probe keys
    float* D;                // the query's row
    int64_t* I;
    int k, nprobe, store_pairs;
};
__device__ __noinline__ void stream_merge_emit(const u64* mbp, u64* mqueue, const uint32_t* cum, const int64_t* poff,
                                               StreamOut o, int lane) {
    WaveSelect<1> ms;
    ms.init(o.k, mqueue, lane);
    for (int w = 0; w < 4; w++)
        for (int e0 = 0; e0 < o.k; e0 += 64) {
            const int e = e0 + lane;
            const bool valid = e < o.k;
            ms.offer_key(valid ? mbp[w * o.k + e] : kMaxKey, valid);
        }
    ms.flush();
    if (lane >= o.k) return;
    const u64 key = ms.best[0];
    float dis = 3.402823466e+38f;          // Heap.h:318-321 padding
    int64_t id = -1;
    if (key != kMaxKey) {
        dis = ordered_to_f32((uint32_t)(key >> 32));
        const uint32_t pos = (uint32_t)key;
        int lo = 0, hi = o.nprobe;         // last probe p with cum[p] <= pos
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] <= pos) lo = mid; else hi = mid;
        }
        const int64_t off = pos - cum[lo];
        id = o.store_pairs ? (o.kq[lo] << 32 | off) : o.ids[poff[lo] + off];   // IndexIVFPQ.cpp:798
    }
    o.D[lane] = dis;
    o.I[lane] = id;
}

}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void scan16s_kernel(ScanArgs a, int lut_region, int pmb, int overlap) {
    constexpr int E = 4096, NT = 256, NI = 4, NW = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                         // [2][E]
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);              // [NW][64]
    u64* mqueue = queue + NW * 64;                                        // [64] queue of the merging wave
    u64* mb = mqueue + 64;                                                // [2][NW][k] parked per-wave lists
    unsigned char* pmbase = reinterpret_cast<unsigned char*>(mb + 2 * NW * a.k);
    const int np2 = (a.nprobe + 1) & ~1;
    uint16_t* ordv = reinterpret_cast<uint16_t*>(pmbase + 3 * pmb);       // [3][np2]
    int32_t* ctl = reinterpret_cast<int32_t*>(ordv + 3 * np2);            // [0..2] nlive, [3..5] wg_thr, [6] claimed slot
    auto PM = [&](int P) { ProbeMeta m; m.carve(pmbase + P * pmb, a.nprobe); return m; };

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
    const int x = (int)(blockIdx.x & 7);
    const int64_t base = (int64_t)x * a.xcd_chunk;
    const int limit = (int)std::max<int64_t>(0, std::min<int64_t>(a.xcd_chunk, a.nq - base));
    auto unit_q = [&](int slot) -> int64_t {
        const int64_t s = base + slot;
        return a.qorder ? (int64_t)a.qorder[s] : s;
    };
    // walking order of a staged query (one wave): prefix sums, max_codes cut, live probes
    auto stage_order = [&](int P) {
        ProbeMeta pm = PM(P);
        const int cut = probe_meta_scan(a, pm, lane);
        __builtin_amdgcn_wave_barrier();
        uint16_t* ord = ordv + P * np2;
        int nl = 0;
        for (int p0 = 0; p0 < cut; p0 += 64) {
            const int p = p0 + lane;
            const bool lv = p < cut && pm.pkey[p] >= 0;
            const u64 mask = __ballot(lv);
            if (lv) ord[nl + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)p;
            nl += __popcll(mask);
        }
        if (lane == 0) { ctl[P] = nl; ctl[3 + P] = (int32_t)f32_to_ordered(3.402823466e+38f); }
    };

    // ---- first query of this workgroup: the plain, serial set-up -------------------------------
    if (t == 0) ctl[6] = atomicAdd(&a.own_next[x], 1);
    __syncthreads();
    int slot = __builtin_amdgcn_readfirstlane(ctl[6]);
    if (slot >= limit) return;
    int64_t q_cur = unit_q(slot);
    int P = 0;
    bool badkey;
    { ProbeMeta pm0 = PM(0); badkey = probe_meta_fill(a, q_cur, pm0, t, NT); }
    float4 m2t3[NI];
    load_query_table16<NI>(a, q_cur, t, lane, wave, m2t3);
    __syncthreads();
    if (wave == 0) stage_order(0);
    __syncthreads();

    WaveSelect<1> sel;
    sel.init(a.k, queue + wave * 64, lane);

    float4 t2r[NI];
    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = make_uint4(0, 0, 0, 0);
    uint32_t n_len = 0, n_pos0 = 0;
    float n_dis0 = 0.f;
    int64_t n_off = 0;
    auto prefetch = [&](int Pq, int i) {     // i-th live probe of the query staged in slot Pq
        const ProbeMeta pm = PM(Pq);
        const int p = ordv[Pq * np2 + i];
        const int64_t key = pm.pkey[p];
        n_len = __builtin_amdgcn_readfirstlane(pm.plen[p]);
        n_dis0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pm.pd0[p])));
        n_pos0 = __builtin_amdgcn_readfirstlane(pm.cum[p]);
        {
            const int64_t o = pm.poff[p];
            n_off = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)o >> 32)) << 32) |
                              __builtin_amdgcn_readfirstlane((uint32_t)o));
        }
        const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)key * E);
#pragma unroll
        for (int i2 = 0; i2 < NI; i2++) t2r[i2] = src[i2 * NT + t];
        const uint4* cpn = reinterpret_cast<const uint4*>(a.codes) + n_off;
        const uint32_t last = n_len - 1;
        c0 = cpn[min((uint32_t)t, last)];
        c1 = cpn[min((uint32_t)t + NT, last)];
    };
    if (ctl[0] > 0) prefetch(0, 0);
    int buf = 0;
    uint64_t nscan = 0;
    int u = 0;                      // queries finished by this workgroup
    bool have_pend = false;         // a finished query waits for its merge
    int pend_P = 0;
    int64_t pend_q = 0;
    auto merge_prev = [&]() {
        const ProbeMeta pmp = PM(pend_P);
        StreamOut o;
        o.ids = a.ids; o.kq = a.keys + pend_q * a.nprobe; o.D = a.D + pend_q * a.k; o.I = a.I + pend_q * a.k;
        o.k = a.k; o.nprobe = a.nprobe; o.store_pairs = a.store_pairs;
        stream_merge_emit(mb + (size_t)((u - 1) & 1) * NW * a.k, mqueue, pmp.cum, pmp.poff, o, lane);
    };
    for (;;) {
        const int nlive = __builtin_amdgcn_readfirstlane(ctl[P]);
        const int Pn = P == 2 ? 0 : P + 1;
        const int e = nlive - 1;
        const bool ov = overlap && nlive >= 6;
        int claimed = 0;
        bool next_ok = false;
        int64_t q_next = -1;
        uint32_t* wg_thr = reinterpret_cast<uint32_t*>(ctl + 3 + P);
        if (nlive == 0 && have_pend) {      // nothing to scan here: the previous query's merge has no probe to ride on
            __syncthreads();
            if (wave == 0) merge_prev();
            have_pend = false;
        }
        for (int i = 0; i < nlive; i++) {
            const uint32_t len = n_len;
            const float dis0 = n_dis0;
            const uint32_t pos0 = n_pos0;
            const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + n_off;
            build_lut16<NI>(lut + buf * E, t, t2r, m2t3);
            uint4 cc = c0, cd = c1;
            if (ov) {
                if (i == e - 5 && t == 0) claimed = atomicAdd(&a.own_next[x], 1);
                if (i == e - 4 && t == 0) ctl[6] = claimed;
                if (i == e - 3) {
                    const int sn = __builtin_amdgcn_readfirstlane(ctl[6]);
                    next_ok = sn < limit;
                    if (next_ok) {
                        q_next = unit_q(sn);
                        ProbeMeta pmn = PM(Pn);
                        badkey = probe_meta_fill(a, q_next, pmn, t, NT) || badkey;
                    }
                }
            }
            if (i + 1 < nlive) prefetch(P, i + 1);
            else if (next_ok) {
                // last probe of this query: the next query's first row and codes take the prefetch slot, and
                // its per-query table part goes into the registers the table build above has just released
                if (ctl[Pn] > 0) prefetch(Pn, 0);
                load_query_table16<NI>(a, q_next, t, lane, wave, m2t3);
            }
            __syncthreads();
            if (ov && i == e - 3 && next_ok && wave == 0) stage_order(Pn);
            if (have_pend && i == 0) {
                if (wave == (u & 3)) merge_prev();
                have_pend = false;
            }
            if (sel.dirty) {
                if (lane == 0) atomicMin(wg_thr, f32_to_ordered(sel.thr_own));
                sel.dirty = false;
            }
            sel.refresh_with(*wg_thr);
            auto scan_list = [&](auto bufc) {
                constexpr int B = decltype(bufc)::value;
                uint32_t j0 = (uint32_t)wave * 64;
                for (; j0 + NT < len; j0 += 2 * NT) {
                    const uint32_t ja = j0 + lane, jb = ja + NT;
                    const uint4 ca = cc, cb = cd;
                    cc = cp[min(jb + NT, len - 1)];
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
            if (buf == 0) scan_list(std::integral_constant<int, 0>{});
            else scan_list(std::integral_constant<int, 1>{});
            nscan += len;
            buf ^= 1;
        }
        // ---- this query is scanned: park the wave's list, start a fresh selection -------------------
        sel.flush();
        {
            u64* mine = mb + ((size_t)(u & 1) * NW + wave) * a.k;
            if (lane < a.k) mine[lane] = sel.best[0];
        }
        sel.init(a.k, queue + wave * 64, lane);
        pend_P = P;
        pend_q = q_cur;
        have_pend = true;
        u++;
        if (next_ok) {              // staged and already prefetched: straight on
            P = Pn;
            q_cur = q_next;
            continue;
        }
        // serial path: a short query (no room to stage the next one) or an exhausted queue
        if (!ov && t == 0) ctl[6] = atomicAdd(&a.own_next[x], 1);
        __syncthreads();            // parked lists (and the claim) are visible
        if (wave == 0) merge_prev();
        have_pend = false;
        if (ov) break;              // the queue was empty when this query looked
        slot = __builtin_amdgcn_readfirstlane(ctl[6]);
        if (slot >= limit) break;
        q_cur = unit_q(slot);
        P = Pn;
        { ProbeMeta pmn = PM(P); badkey = probe_meta_fill(a, q_cur, pmn, t, NT) || badkey; }
        load_query_table16<NI>(a, q_cur, t, lane, wave, m2t3);
        __syncthreads();
        if (wave == 0) stage_order(P);
        __syncthreads();
        if (ctl[P] > 0) prefetch(P, 0);
    }
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
    if (badkey) *a.bad_key = 1;
}

void launch_scan16_stream(const ScanArgs& a_in, hipStream_t s, int overlap) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    a.nsplit = 1;
    a.xcd_chunk = (int)((a.nq + 7) / 8);
    const size_t lutb = (size_t)2 * 4096 * 4;
    const int pmb = (int)(((size_t)a.nprobe * 24 + 8 + 15) & ~(size_t)15);
    const size_t smem = lutb + (size_t)(4 * 64 + 64) * 8 + (size_t)2 * 4 * a.k * 8 + (size_t)3 * pmb +
                        (size_t)3 * ((a.nprobe + 1) & ~1) * 2 + 64;
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan16s_kernel), smem);
    (void)hipMemsetAsync(a.own_next, 0, 8 * sizeof(int), s);
    const unsigned per_xcd = (unsigned)std::min<int64_t>(128, a.xcd_chunk);
    hipLaunchKernelGGL(scan16s_kernel, dim3(8 * per_xcd), dim3(256), smem, s, a, (int)lutb, pmb, overlap);
}

}  // namespace vlq
