// Short-list scan for the other code sizes (round 6): scan16_short_kernel's organisation over M.
//
// An index of many lists holds a few codes per list (the inverted multi-index of the reference's own drivers: 2^28 lists;
// tests/sift1b_imi_pq.cpp and tests/deep1b_imi_pq.cpp ship 8-BYTE codes).  Building the M x 256-entry table of a probed list
// (precompute_list_tables_L2, IndexIVFPQ.cpp:631-690) to look up M x a-handful of entries is what bounds the engineered
// kernels of scanm.hip there, so sparse indexes of every size but 16 bytes fell through to the generic kernel: 8-byte codes on
// 2^28 lists, 250 M vectors, 10 000 queries: 2.10 ms of scan against 0.40 ms for 16-byte codes -- five times the time for half
// the bytes.  Here, as in scan16_short_kernel: no table is built; the per-query part -2 <q_m, cent_mj> sits in LDS once per
// query (from the per-query table the generic path already computes: any sub-vector width), each wave walks its own probes
// (no workgroup barrier in the loop), and a lane fetches exactly the M entries of term 2 its code addresses and forms the SAME
// table entries term2 + (-2 <q, cent>) (fvec_madd, IndexIVFPQ.cpp:641-644) before the left-to-right sum: identical arithmetic,
// identical results (tests/test_gpu_imi_wide.py::test_sparse_lists_of_the_other_code_sizes: eight shapes incl. a flat quantizer,
// store_pairs and ncode, against the oracle bit for bit; the reference drivers' runs: tests/test_reference_drivers.py).
// Multi-index cells are walked in (first half, second half) order like the 16-byte kernel's (profiles/r06_scan16_short_pmc.txt).
#include "scan16_common.cuh"
#include "scan_common.cuh"

namespace vlq {

namespace {
template <int M>
__device__ __forceinline__ void load_code_words(const uint8_t* p, uint32_t (&w)[M / 4]) {
    if constexpr (M % 16 == 0) {
#pragma unroll
        for (int i = 0; i < M / 16; i++) {
            const uint4 v = reinterpret_cast<const uint4*>(p)[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    } else if constexpr (M % 8 == 0) {
#pragma unroll
        for (int i = 0; i < M / 8; i++) {
            const uint2 v = reinterpret_cast<const uint2*>(p)[i];
            w[2 * i] = v.x; w[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < M / 4; i++) w[i] = reinterpret_cast<const uint32_t*>(p)[i];
    }
}
}  // namespace

template <int M, int KPL>
__global__ __launch_bounds__(256) void scanm_short_kernel(ScanArgs a, int queue_off) {
    constexpr int E = M * 256, NW = 4, NT = 256, NWORD = M / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* qtl = reinterpret_cast<float*>(smraw);                         // [M][256] -2 <q_m, cent_mj>
    u64* queue = reinterpret_cast<u64*>(smraw + queue_off);               // [NW][64]
    ProbeMeta pm;
    pm.carve(reinterpret_cast<unsigned char*>(queue + NW * 64), a.nprobe);
    int32_t* misc = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(queue + NW * 64) + ProbeMeta::bytes(a.nprobe));
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
        const float4* qt = reinterpret_cast<const float4*>(a.qtab + q * E);
        for (int i = t; i < E / 4; i += NT) {
            const float4 v = qt[i];
            reinterpret_cast<float4*>(qtl)[i] = a.qtab_scaled ? v : make_float4(__fmul_rn(-2.f, v.x), __fmul_rn(-2.f, v.y), __fmul_rn(-2.f, v.z),
                                                                                __fmul_rn(-2.f, v.w));
        }
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
        if (a.imi_nbits > 0 && nl <= 64 && nl > 1 && !a.short_keep_order) {      // cells by halves (see scan16_short_kernel)
            __builtin_amdgcn_wave_barrier();
            const int p = lane < nl ? ord[lane] : 0;
            const int64_t key = lane < nl ? (int64_t)pm.pkey[p] : 0;
            const u64 i0 = (u64)(key & ((int64_t(1) << a.imi_nbits) - 1)), i1 = (u64)(key >> a.imi_nbits);
            const u64 so = wave_sort64(lane < nl ? ((i0 << 40) | (i1 << 16) | (u64)p) : kMaxKey, lane);
            __builtin_amdgcn_wave_barrier();
            if (lane < nl) ord[lane] = (uint16_t)(so & 0xffffu);
        }
        if (lane == 0) { misc[0] = cut; misc[1] = nl; }
    }
    __syncthreads();
    const int nlive = misc[1];

    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);
    auto code_at = [&](int p, uint32_t j, uint32_t (&w)[NWORD]) {
        load_code_words<M>(a.codes + (size_t)(pm.poff[p] + (int64_t)min(j, pm.plen[p] - 1)) * M, w);
    };
    uint32_t wn[NWORD];
#pragma unroll
    for (int i = 0; i < NWORD; i++) wn[i] = 0;
    if (wave < nlive) code_at(ord[wave], (uint32_t)lane, wn);
    for (int i = wave; i < nlive; i += NW) {          // this wave's probes
        const int p = ord[i];
        const uint32_t len = pm.plen[p];
        const float dis0 = pm.pd0[p];
        const uint32_t pos0 = pm.cum[p];
        const int64_t key = pm.pkey[p];
        const float* row0;
        const float* row1;
        if (a.imi_nbits > 0) {      // table type 2 (IndexIVFPQ.cpp:645-686): halves from two rows
            row0 = a.term2 + (size_t)(key & ((int64_t(1) << a.imi_nbits) - 1)) * E;
            row1 = a.term2 + (size_t)(key >> a.imi_nbits) * E;
        } else {
            row0 = row1 = a.term2 + (size_t)key * E;
        }
        uint32_t w[NWORD];
#pragma unroll
        for (int x = 0; x < NWORD; x++) w[x] = wn[x];
        if (i + NW < nlive) code_at(ord[i + NW], (uint32_t)lane, wn);       // the next probe's first codes under this probe's sums
        for (uint32_t j0 = 0; j0 < len; j0 += 64) {
            const uint32_t j = j0 + lane;
            uint32_t wf[NWORD];
            const bool more = j0 + 64 < len;
            if (more) code_at(p, j + 64, wf);
            float e[M];
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t c = (w[m >> 2] >> (8 * (m & 3))) & 255u;
                const float t2 = (m < M / 2 ? row0 : row1)[m * 256 + c];
                e[m] = __fadd_rn(t2, qtl[m * 256 + c]);
            }
            float dis = dis0;
#pragma unroll
            for (int m = 0; m < M; m++) dis = __fadd_rn(dis, e[m]);
            sel.offer_keyed(dis, pos0 + j, j < len);
            if (more) {
#pragma unroll
                for (int x = 0; x < NWORD; x++) w[x] = wf[x];
            }
        }
    }
    merge_and_emit<KPL, NW>(sel, smraw, pm.cum, a, q, wave, lane,
                            [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)pm.cum[a.nprobe]);
    if (badkey) *a.bad_key = 1;
}

bool scanm_short_supports(const ScanArgs& a) {
    const bool size_ok = a.M == 4 || a.M == 8 || a.M == 12 || (a.M >= 20 && a.M <= 32 && a.M % 4 == 0) || (a.M >= 40 && a.M <= 64 && a.M % 8 == 0);
    return size_ok && a.ksub == 256 && a.table_mode == 1 && a.term2 && a.qtab && a.nprobe <= 1024 && a.k >= 1 && a.k <= 1024 && a.nsplit == 1 &&
           a.tail_r == 0;
}

template <int M, int KPL>
static void launch_scanm_short_t(const ScanArgs& a, int queue_off, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(scanm_short_kernel<M, KPL>), smem);
    hipLaunchKernelGGL((scanm_short_kernel<M, KPL>), dim3((unsigned)(8 * a.xcd_chunk)), dim3(256), smem, s, a, queue_off);
}

template <int M>
static void launch_scanm_short_m(const ScanArgs& a, hipStream_t s) {
    size_t region = (size_t)M * 256 * 4;                    // the merge area aliases the table
    const size_t merge = (size_t)4 * a.k * 8;
    if (region < merge) region = merge;
    const size_t smem = region + (size_t)4 * 64 * 8 + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 2 + 64;
    if (a.k <= 64) launch_scanm_short_t<M, 1>(a, (int)region, smem, s);
    else if (a.k <= 256) launch_scanm_short_t<M, 4>(a, (int)region, smem, s);
    else launch_scanm_short_t<M, 16>(a, (int)region, smem, s);
}

void launch_scanm_short(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    a.xcd_chunk = (int)((a.nq + 7) / 8);
    static const bool keep_order = getenv("VLQ_SHORT_KEEP_ORDER") != nullptr;
    a.short_keep_order = keep_order ? 1 : 0;
    switch (a.M) {
    case 4: launch_scanm_short_m<4>(a, s); break;
    case 8: launch_scanm_short_m<8>(a, s); break;
    case 12: launch_scanm_short_m<12>(a, s); break;
    case 20: launch_scanm_short_m<20>(a, s); break;
    case 24: launch_scanm_short_m<24>(a, s); break;
    case 28: launch_scanm_short_m<28>(a, s); break;
    case 32: launch_scanm_short_m<32>(a, s); break;
    case 40: launch_scanm_short_m<40>(a, s); break;
    case 48: launch_scanm_short_m<48>(a, s); break;
    case 56: launch_scanm_short_m<56>(a, s); break;
    default: launch_scanm_short_m<64>(a, s); break;
    }
}

}  // namespace vlq
