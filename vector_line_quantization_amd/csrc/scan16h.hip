// 16-byte-code list scan with FLOAT16 look-up tables -- GpuIndexIVFPQConfig::useFloat16LookupTables on the plain
// IVFPQ path (gpu/GpuIndexIVFPQ.h:24-38; the reference's pqScanPrecomputedMultiPass with LookupT = half,
// gpu/impl/PQScanMultiPassPrecomputed.cu:30-114,375-477; exercised by gpu/test/TestGpuIndexIVFPQ.cpp:44-69).
// As in the reference: term 2 and term 3 are kept as half (impl/IVFPQ.cu:599-684, :1599-1680 toHalf), a list's table
// is their HALF sum (loadPrecomputedTerm, Math<Half8>::add), and a code's distance is term 1 plus the looked-up
// entries converted to float, added left to right (:431-449).  Opt-in (vlq_ivfpq_set_float16_tables): the fp32
// kernels, which are the parity build against the CPU library, are untouched.  Bit-identical to the oracle's
// float16 mode (oracle/ivfpq_oracle.cpp, float16_tables).
//
// What it buys: a probe's row is 8 KB instead of 16 KB -- on data whose queries share few probes the fp32 scan is
// bound by exactly those rows coming from the Infinity Cache ("data sensitivity").  Same organisation as
// scan16_kernel otherwise: one workgroup per query in the XCD-aware order, probe metadata in LDS, the next probe's
// row and first codes prefetched, double-buffered table, one barrier per probe, wave-level running selection.
#include "kernels.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "walk_order.cuh"
#include "wave_topk.cuh"

namespace vlq {

typedef _Float16 h16x2s __attribute__((ext_vector_type(2)));
union H8s { uint4 u; h16x2s h[4]; };

// 8 code bytes (words W0, W1) -> 8 half entries of the table at LDS byte O (+ m * 512); one SDWA op makes byte * 2
#define VLQ_S16H_BLOCK(W0, W1, O)                                                                                  \
    asm volatile(                                                                                                  \
        "v_lshlrev_b32_sdwa %0, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"    \
        "v_lshlrev_b32_sdwa %1, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"    \
        "v_lshlrev_b32_sdwa %2, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"    \
        "v_lshlrev_b32_sdwa %3, %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"    \
        "v_lshlrev_b32_sdwa %4, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"    \
        "v_lshlrev_b32_sdwa %5, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"    \
        "v_lshlrev_b32_sdwa %6, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"    \
        "v_lshlrev_b32_sdwa %7, %10, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"    \
        "ds_read_u16 %0, %0 offset:" #O "+0\n\t"                                                                   \
        "ds_read_u16 %1, %1 offset:" #O "+512\n\t"                                                                 \
        "ds_read_u16 %2, %2 offset:" #O "+1024\n\t"                                                                \
        "ds_read_u16 %3, %3 offset:" #O "+1536\n\t"                                                                \
        "ds_read_u16 %4, %4 offset:" #O "+2048\n\t"                                                                \
        "ds_read_u16 %5, %5 offset:" #O "+2560\n\t"                                                                \
        "ds_read_u16 %6, %6 offset:" #O "+3072\n\t"                                                                \
        "ds_read_u16 %7, %7 offset:" #O "+3584\n\t"                                                                \
        "s_waitcnt lgkmcnt(0)"                                                                                     \
        : "=&v"(vh[0]), "=&v"(vh[1]), "=&v"(vh[2]), "=&v"(vh[3]), "=&v"(vh[4]), "=&v"(vh[5]), "=&v"(vh[6]),        \
          "=&v"(vh[7])                                                                                             \
        : "v"(W0), "v"(W1), "v"(one)                                                                               \
        : "memory")

template <int KPL>
__global__ __launch_bounds__(256) void scan16h_kernel(ScanArgs a, int lut_region) {
    constexpr int E = 4096, NT = 256, NI = 2;      // a row of 4096 halves = 512 x 16 bytes: two per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    // LDS: [2 tables of 8 KB at bytes 0 and 8192][queue 4 x 64 keys][probe metadata][misc][ord]
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);              // [4][64]
    ProbeMeta pm;
    pm.carve(reinterpret_cast<unsigned char*>(queue + 4 * 64), a.nprobe);
    int32_t* misc = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(queue + 4 * 64) + ProbeMeta::bytes(a.nprobe));
    uint16_t* ord = reinterpret_cast<uint16_t*>(misc + 2);                // [nprobe] visited probes, in walking order

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }     // the table offsets are absolute
    uint32_t one = 1;
    asm volatile("" : "+v"(one));
    // XCD-aware placement of the sorted query order, as scan16_kernel
    const int64_t b = blockIdx.x;
    const int64_t s = (b & 7) * a.xcd_chunk + (b >> 3);
    if (s >= a.nq) return;
    const int64_t q = a.qorder ? a.qorder[s] : s;
    const int64_t* kq = a.keys + q * a.nprobe;

    const bool badkey = probe_meta_fill(a, q, pm, t, NT);
    H8s q3[NI];                           // half(-2 <q_m, cent_mj>), entries 8 * (i * 256 + t) .. + 7
    {
        const uint4* qt = reinterpret_cast<const uint4*>(a.qtabh + q * E);
#pragma unroll
        for (int i = 0; i < NI; i++) q3[i].u = qt[i * NT + t];
    }
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
        walk_order_sort(a, pm, ord, nl, lane);
        if (lane == 0) { misc[0] = cut; misc[1] = nl; }
    }
    __syncthreads();
    const int nlive = misc[1];

    WaveSelect<KPL, 1, (KPL >= 2)> sel;
    sel.init(a.k, queue + wave * 64, lane);

    H8s t2r[NI];
    uint4 c0 = make_uint4(0, 0, 0, 0);
    uint32_t n_len = 0, n_pos0 = 0;
    float n_dis0 = 0.f;
    int64_t n_off = 0;
    auto prefetch = [&](int i) {
        if (i >= nlive) return;
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
        const uint4* src = reinterpret_cast<const uint4*>(a.term2h + (size_t)key * E);
#pragma unroll
        for (int i2 = 0; i2 < NI; i2++) t2r[i2].u = src[i2 * NT + t];
        c0 = (reinterpret_cast<const uint4*>(a.codes) + n_off)[min((uint32_t)t, n_len - 1)];
    };
    prefetch(0);
    int buf = 0;
    uint64_t nscan = 0;
    for (int i = 0; i < nlive; i++) {
        const uint32_t len = n_len;
        const float dis0 = n_dis0;
        const uint32_t pos0 = n_pos0;
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + n_off;
        uint4* L = reinterpret_cast<uint4*>(smraw + buf * 8192);
#pragma unroll
        for (int i2 = 0; i2 < NI; i2++) {         // term23h = hadd(term2h[list], term3h[q]), round to nearest even
            H8s v;
#pragma unroll
            for (int e = 0; e < 4; e++) v.h[e] = t2r[i2].h[e] + q3[i2].h[e];
            L[i2 * NT + t] = v.u;
        }
        uint4 cc = c0;
        prefetch(i + 1);
        __syncthreads();
        auto scan_list = [&](auto bufc) {
            constexpr int B = decltype(bufc)::value;
            auto h2f = [](uint32_t v) { return (float)__builtin_bit_cast(_Float16, (uint16_t)v); };
            for (uint32_t j0 = (uint32_t)wave * 64; j0 < len; j0 += NT) {
                const uint32_t j = j0 + lane;
                const uint4 cn = cp[min(j + NT, len - 1)];
                float dis = dis0;
                {
                    uint32_t vh[8];
                    if (B == 0) VLQ_S16H_BLOCK(cc.x, cc.y, 0); else VLQ_S16H_BLOCK(cc.x, cc.y, 8192);
#pragma unroll
                    for (int m = 0; m < 8; m++) dis = __fadd_rn(dis, h2f(vh[m]));
                }
                {
                    uint32_t vh[8];
                    if (B == 0) VLQ_S16H_BLOCK(cc.z, cc.w, 4096); else VLQ_S16H_BLOCK(cc.z, cc.w, 12288);
#pragma unroll
                    for (int m = 0; m < 8; m++) dis = __fadd_rn(dis, h2f(vh[m]));
                }
                sel.offer_keyed(dis, pos0 + j, j < len);
                cc = cn;
            }
        };
        if (buf == 0) scan_list(std::integral_constant<int, 0>{});
        else scan_list(std::integral_constant<int, 1>{});
        nscan += len;
        buf ^= 1;
    }
    merge_and_emit<KPL, 4, 1>(sel, smraw, pm.cum, a, q, wave, lane,
                              [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = pm.poff[p]; });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
    if (badkey) *a.bad_key = 1;
}

template <int KPL>
static void launch_scan16h_t(const ScanArgs& a, int lut_region, size_t smem, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan16h_kernel<KPL>), smem);
    const unsigned grid = (unsigned)(((a.nq + 7) / 8) * 8);
    ScanArgs b = a;
    b.xcd_chunk = (a.nq + 7) / 8;
    hipLaunchKernelGGL((scan16h_kernel<KPL>), dim3(grid), dim3(256), smem, s, b, lut_region);
}

bool scan16h_supports(const ScanArgs& a) {
    return a.M == 16 && a.ksub == 256 && a.table_mode == 1 && a.imi_nbits == 0 && a.k <= 256 && a.term2h && a.qtabh;
}

void launch_scan16h(const ScanArgs& a, hipStream_t s) {
    if (a.nq <= 0) return;
    size_t lutb = (size_t)2 * 8192;
    const size_t merge = (size_t)4 * a.k * 8;
    if (lutb < merge) lutb = merge;
    const size_t smem = lutb + 4 * 64 * 8 + (size_t)a.nprobe * 24 + 8 + 8 + (size_t)a.nprobe * 2 + 8 + 64;
    if (a.k <= 64) launch_scan16h_t<1>(a, (int)lutb, smem, s);
    else if (a.k <= 128) launch_scan16h_t<2>(a, (int)lutb, smem, s);
    else launch_scan16h_t<4>(a, (int)lutb, smem, s);
}

// largest |x[i]| (as the bit pattern of a non-negative float: orders like an integer)
__global__ void max_abs_kernel(const float* __restrict__ x, int64_t n, unsigned int* __restrict__ out) {
    unsigned int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int u = __float_as_uint(x[i]) & 0x7fffffffu;
        m = u > m ? u : m;
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) { const unsigned int o = __shfl_xor(m, sft, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
void launch_max_abs(const float* x, int64_t n, unsigned int* out, hipStream_t s) {
    (void)hipMemsetAsync(out, 0, sizeof(unsigned int), s);
    if (n <= 0) return;
    hipLaunchKernelGGL(max_abs_kernel, dim3(2048), dim3(256), 0, s, x, n, out);
}

}  // namespace vlq
