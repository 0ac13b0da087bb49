// List-scan kernel for 16-byte codes (M = 16, ksub = 256, table mode 1 / 2), second
// generation of scan16.hip.  Same arithmetic (IndexIVFPQ.cpp:631-690, :781-802), same
// workgroup shape (one 256-thread workgroup per query, double-buffered 2 x 16 KB LUT, one
// barrier per probe); what changed is the memory pipeline and the instruction count:
//   * vmcnt retires loads IN ORDER, so a load that is waited on drags every older load
//     with it.  scan16.hip requested the next probe's term2 row and then, inside the list
//     loop, the next code chunk: the first in-loop wait therefore also waited for the
//     "one probe ahead" prefetch, which in effect ran one ITERATION ahead.  Here a wave's
//     first PD code chunks of a list sit in registers before the list is scanned, and the
//     only loads issued while scanning probe p are the ones probe p+1 needs (its term2
//     row right after the LUT build, its code chunks into each chunk register as soon as
//     that register has been consumed): nothing scanning probe p waits for them, they
//     have a whole probe to land.  Lists longer than PD chunks per wave fall back to
//     PD-deep in-list prefetch for the excess.
//   * list bases and lengths are scalars (readfirstlane of the LDS probe metadata): all
//     loads are SGPR base + 32-bit VGPR offset, no 64-bit vector address arithmetic;
//   * the running selection is offered once per PD chunks through a single call site
//     (the common case -- no lane beats the threshold -- is PD compares and one branch).
// Results are bit-identical to scan16.hip: every wave still sees its candidates in
// increasing scan position, which is what the strict `<` admission needs.
#include <type_traits>

#include "kernels.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "wave_topk.cuh"

namespace vlq {

__device__ __forceinline__ uint32_t sread(const uint32_t* p) { return __builtin_amdgcn_readfirstlane(*p); }
__device__ __forceinline__ uint64_t sread64(const int64_t* p) {
    const uint64_t v = (uint64_t)*p;
    return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
           __builtin_amdgcn_readfirstlane((uint32_t)v);
}
__device__ __forceinline__ uint4 ld16(const unsigned char* sbase, uint32_t voff) {
    return *reinterpret_cast<const uint4*>(sbase + voff);
}

template <int KPL>
__global__ __launch_bounds__(256, KPL <= 4 ? 4 : 3) void scan16v2_kernel(ScanArgs a, int lut_region) {
    constexpr int E = 4096;
    constexpr int NW = 4, NT = 256, NI = 4;
    constexpr int PD = 4;             // code chunks (64 codes each) a wave holds in registers
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                         // [2][E]
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);              // [NW][64]
    ProbeMeta pm;
    pm.carve(reinterpret_cast<unsigned char*>(queue + NW * 64), a.nprobe);
    int32_t* misc = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(queue + NW * 64) +
                                               ProbeMeta::bytes(a.nprobe));   // cut, nlive
    uint32_t* live = reinterpret_cast<uint32_t*>(misc + 2);               // [nprobe] visited probes, in order

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }   // adc16_fixed: LUTs at LDS 0 / 16384
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
    int64_t q;
    {
        const int64_t b = blockIdx.x;
        const int64_t s = (b & 7) * a.xcd_chunk + (b >> 3);
        if (s >= a.nq) return;
        q = a.qorder ? a.qorder[s] : s;
    }
    const int64_t* kq = a.keys + q * a.nprobe;

    // ---- per-query set-up ---------------------------------------------------------
    const bool badkey = probe_meta_fill(a, q, pm, t, NT);
    float4 m2t3[NI];
    load_query_table16<NI>(a, q, t, lane, wave, m2t3);
    __syncthreads();
    if (wave == 0) {
        const int cut = probe_meta_scan(a, pm, lane);
        __builtin_amdgcn_wave_barrier();
        int nl = 0;
        for (int p0 = 0; p0 < cut; p0 += 64) {
            const int p = p0 + lane;
            const bool lv = p < cut && pm.pkey[p] >= 0;
            const u64 mask = __ballot(lv);
            if (lv) live[nl + __popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)p;
            nl += __popcll(mask);
        }
        if (lane == 0) { misc[0] = cut; misc[1] = nl; }
    }
    __syncthreads();
    const int nlive = __builtin_amdgcn_readfirstlane(misc[1]);

    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);

    // per-thread constants: byte offset of this thread's LUT slices / of its code in chunk 0
    uint32_t voff_t2[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) voff_t2[i] = (uint32_t)(i * NT + t) * 16u;
    const uint32_t lane_off = (uint32_t)(wave * 64 + lane) * 16u;
    const unsigned char* codes = reinterpret_cast<const unsigned char*>(a.codes);
    const unsigned char* term2 = reinterpret_cast<const unsigned char*>(a.term2);

    struct ListRef {                    // all wave-uniform (SGPRs)
        const unsigned char* base;      // first code of the list
        uint32_t lenb;                  // list length in bytes (len * 16)
        const unsigned char* row0;      // term2 row (sub-quantizers 0-7) ...
        const unsigned char* row1;      // ... and 8-15 (differs from row0 only for table type 2)
    };
    auto list_ref = [&](int li) __attribute__((always_inline)) {
        ListRef r;
        const uint32_t p = sread(&live[li]);
        const uint32_t key = sread(reinterpret_cast<const uint32_t*>(&pm.pkey[p]));
        r.base = codes + sread64(&pm.poff[p]) * 16u;
        r.lenb = sread(&pm.plen[p]) * 16u;
        if (a.imi_nbits > 0) {          // table type 2 (IndexIVFPQ.cpp:645-686)
            const uint32_t ki0 = key & ((1u << a.imi_nbits) - 1u), ki1 = key >> a.imi_nbits;
            r.row0 = term2 + (size_t)ki0 * (E * 4);
            r.row1 = term2 + (size_t)ki1 * (E * 4);
        } else {
            r.row0 = r.row1 = term2 + (size_t)key * (E * 4);
        }
        return r;
    };
    float4 t2r[NI];
    uint4 C[PD];
    auto load_row = [&](const ListRef& r) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const uint4 v = ld16((NW * i + wave) < 8 ? r.row0 : r.row1, voff_t2[i]);
            t2r[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };

    if (nlive > 0) {
        const ListRef r0 = list_ref(0);
        load_row(r0);
#pragma unroll
        for (int u = 0; u < PD; u++) C[u] = ld16(r0.base, min(lane_off + (uint32_t)u * 4096u, r0.lenb - 16u));
    }
    // two probes per trip (LUT buffer 0, then 1): straight-line code, so the chunk registers
    // keep their places across the back edge; an odd count is padded by revisiting the
    // last list with nothing to scan
    auto probe = [&](auto bufc, int li) __attribute__((always_inline)) {
        constexpr int B = decltype(bufc)::value;
        const int lc = min(li, nlive - 1);
        const uint32_t p = sread(&live[lc]);
        const ListRef cur = list_ref(lc);
        const ListRef nxt = list_ref(min(li + 1, nlive - 1));    // last probe: harmless re-reads of itself
        const uint32_t scanb = li < nlive ? cur.lenb : 0u;
        const float dis0 = __uint_as_float(sread(reinterpret_cast<const uint32_t*>(&pm.pd0[p])));
        const uint32_t pos0 = sread(&pm.cum[p]);
        build_lut16<NI>(lut + B * E, t, t2r, m2t3);
        load_row(nxt);
        __syncthreads();
        // chunks of this list that belong to this wave: it = 0 .. nit-1 at byte lane_off + it * 4096
        const uint32_t wb = (uint32_t)wave * 1024u;
        const uint32_t nit = scanb > wb ? (scanb - wb + 4095u) >> 12 : 0u;
        auto group = [&](uint32_t i0) __attribute__((always_inline)) {
            float d[PD];
            u64 okm[PD];
            u64 any = 0;
#pragma unroll
            for (int u = 0; u < PD; u++) d[u] = 3.402823466e+38f;
            // the wave's chunks of this group as ONE straight-line half-block pipeline (scan16_common.cuh)
            const uint32_t left = nit > i0 ? nit - i0 : 0u;              // wave-uniform
            if (left >= 4) {
                adc16_pipeline<4, B * 16384>(C, dis0, two, d);
            } else if (left == 3) {
                const uint4 c3[3] = {C[0], C[1], C[2]};
                float a3[3];
                adc16_pipeline<3, B * 16384>(c3, dis0, two, a3);
                d[0] = a3[0]; d[1] = a3[1]; d[2] = a3[2];
            } else if (left == 2) {
                const uint4 c2[2] = {C[0], C[1]};
                float a2[2];
                adc16_pipeline<2, B * 16384>(c2, dis0, two, a2);
                d[0] = a2[0]; d[1] = a2[1];
            } else if (left == 1) {
                d[0] = adc16_fixed<B>(C[0], dis0, two);
            }
#pragma unroll
            for (int u = 0; u < PD; u++) {
                const uint32_t it = i0 + u;
                const uint32_t off = lane_off + it * 4096u;
                okm[u] = __builtin_amdgcn_ballot_w64(it < nit && off < scanb && d[u] < sel.thr);
                any |= okm[u];
                // the register just consumed gets the chunk this wave needs in it next: PD chunks
                // further down this list, or chunk u of the next list
                const bool same = it + PD < nit;                          // wave-uniform
                const unsigned char* nb = same ? cur.base : nxt.base;
                const uint32_t nlast = (same ? cur.lenb : nxt.lenb) - 16u;
                const uint32_t noff = same ? off + PD * 4096u : lane_off + (uint32_t)u * 4096u;
                C[u] = ld16(nb, min(noff, nlast));
            }
            if (any) {
                // rare: some lane beat the threshold.  One call site for the selection, chunks in
                // increasing scan position
#pragma nounroll
                for (int u = 0; u < PD; u++) {
                    const float du = u == 0 ? d[0] : u == 1 ? d[1] : u == 2 ? d[2] : d[3];
                    const u64 mu = u == 0 ? okm[0] : u == 1 ? okm[1] : u == 2 ? okm[2] : okm[3];
                    sel.offer(du, pos0 + ((lane_off + (i0 + u) * 4096u) >> 4), (mu >> lane) & 1);
                }
            }
        };
        group(0u);                                             // peeled: has its own vmcnt bookkeeping
        for (uint32_t i0 = PD; i0 < nit; i0 += PD) group(i0);
    };
    for (int li = 0; li < nlive; li += 2) {
        probe(std::integral_constant<int, 0>{}, li);
        probe(std::integral_constant<int, 1>{}, li + 1);
    }

    merge_and_emit<KPL, NW>(sel, smraw, pm.cum, a, q, wave, lane,
                            [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)pm.cum[a.nprobe]);
    if (badkey) *a.bad_key = 1;
}

template <int KPL>
static void launch_scan16v2_t(const ScanArgs& a, int lut_region, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan16v2_kernel<KPL>), smem);
    const unsigned grid = (unsigned)(8 * a.xcd_chunk);
    hipLaunchKernelGGL((scan16v2_kernel<KPL>), dim3(grid), dim3(256), smem, s, a, lut_region);
}

void launch_scan16v2(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    a.xcd_chunk = (int)((a.nq + 7) / 8);
    size_t lutb = (size_t)2 * 4096 * 4;
    const size_t merge = (size_t)4 * a.k * 8;
    if (lutb < merge) lutb = merge;
    const size_t tail = (size_t)4 * 64 * 8 + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 4 + 64;
    const size_t smem = lutb + tail;
    if (a.k <= 64) launch_scan16v2_t<1>(a, (int)lutb, smem, s);
    else if (a.k <= 256) launch_scan16v2_t<4>(a, (int)lutb, smem, s);
    else launch_scan16v2_t<16>(a, (int)lutb, smem, s);
}

}  // namespace vlq
