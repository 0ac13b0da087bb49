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
//     mostly L2 hits instead of fabric reads.  Placement only affects speed.
#include "kernels.h"
#include "scan_common.cuh"
#include "wave_topk.cuh"

namespace vlq {

template <int KPL>
__global__ __launch_bounds__(256) void scan16_kernel(ScanArgs a, int lut_region) {
    constexpr int E = 4096;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                         // [2][E]
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);              // [4][64]
    int64_t* poff = reinterpret_cast<int64_t*>(queue + 4 * 64);           // [nprobe] list start
    uint32_t* cum = reinterpret_cast<uint32_t*>(poff + a.nprobe);         // [nprobe+1] scan pos
    uint32_t* plen = cum + a.nprobe + 1;                                  // [nprobe]
    int32_t* pkey = reinterpret_cast<int32_t*>(plen + a.nprobe);          // [nprobe] (-1 = dead)
    float* pd0 = reinterpret_cast<float*>(pkey + a.nprobe);               // [nprobe]
    int32_t* misc = reinterpret_cast<int32_t*>(pd0 + a.nprobe);           // [0] np_eff

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // XCD-aware placement: hardware deals consecutive workgroups round-robin over the 8
    // XCDs, so give XCD x the x-th contiguous chunk of the (sorted) query order.
    int64_t q;
    {
        const int64_t b = blockIdx.x;
        const int64_t s = (b & 7) * a.xcd_chunk + (b >> 3);
        if (s >= a.nq) return;
        q = a.qorder ? a.qorder[s] : s;
    }
    const int64_t* kq = a.keys + q * a.nprobe;
    const float* cq = a.coarse_dis + q * a.nprobe;

    // ---- per-query set-up -------------------------------------------------------
    bool badkey = false;
    for (int p = t; p < a.nprobe; p += 256) {
        const int64_t key = kq[p];
        if (key >= a.nlist) badkey = true;                 // IndexIVFPQ.cpp:1008-1011
        const bool live = key >= 0 && key < a.nlist;
        int64_t off = 0, len = 0;
        if (live) { off = a.list_off[key]; len = a.list_off[key + 1] - off; }
        poff[p] = off;
        plen[p] = (uint32_t)len;
        pkey[p] = (live && len > 0) ? (int32_t)key : -1;   // empty lists are skipped (:1016)
        pd0[p] = cq[p];
    }
    // -2 * sim_table_2 of this query (ProductQuantizer::compute_inner_prod_table,
    // ProductQuantizer.cpp:424-436), 16 entries per thread, kept in registers for all
    // probes.  Entry e = 4*(i*256+t)+c  ->  sub-quantizer m = 4i + wave (wave-uniform),
    // centroid j = 4*lane + c: the four centroids of a thread are 128 contiguous bytes
    // of the L2-resident 128 KB codebook, the query sub-vector is a scalar load.
    float4 m2t3[4];
    if (a.qtab) {
        const float4* qt = reinterpret_cast<const float4*>(a.qtab + q * E);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 v = qt[i * 256 + t];
            m2t3[i] = make_float4(__fmul_rn(-2.f, v.x), __fmul_rn(-2.f, v.y), __fmul_rn(-2.f, v.z),
                                  __fmul_rn(-2.f, v.w));
        }
    } else {
        // codebook read from its transposed copy pq_cent_t[m][component][j]: the four
        // centroids j = 4*lane..4*lane+3 of one component are one 16-byte load, a wave
        // reads 1 KiB contiguous per instruction
        const float* qv = a.queries + q * 128;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = 4 * i + wave;
            const float4* ct = reinterpret_cast<const float4*>(a.pq_cent_t + (size_t)m * 8 * 256) + lane;
            const float4 x0 = *reinterpret_cast<const float4*>(qv + m * 8);
            const float4 x1 = *reinterpret_cast<const float4*>(qv + m * 8 + 4);
            const float4 y0 = ct[0 * 64], y1 = ct[1 * 64], y2 = ct[2 * 64], y3 = ct[3 * 64];
            const float4 y4 = ct[4 * 64], y5 = ct[5 * 64], y6 = ct[6 * 64], y7 = ct[7 * 64];
            // fvec_inner_product, d = 8 (utils.cpp:509-533): s_l = ((0 + x_l y_l) + x_{l+4} y_{l+4}) + 0,
            // result (s0+s1)+(s2+s3); .x/.y/.z/.w = centroids 4*lane+0..3
#define VLQ_IP8(C)                                                                                        \
    __fmul_rn(-2.f,                                                                                      \
              __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.x, y0.C)), __fmul_rn(x1.x, y4.C)), 0.f), \
                                  __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.y, y1.C)), __fmul_rn(x1.y, y5.C)), 0.f)), \
                        __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.z, y2.C)), __fmul_rn(x1.z, y6.C)), 0.f), \
                                  __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(x0.w, y3.C)), __fmul_rn(x1.w, y7.C)), 0.f))))
            m2t3[i] = make_float4(VLQ_IP8(x), VLQ_IP8(y), VLQ_IP8(z), VLQ_IP8(w));
#undef VLQ_IP8
        }
    }
    __syncthreads();
    if (wave == 0) {
        // exclusive prefix sum of the list lengths = scan position of each probe's first
        // code; max_codes cut (IndexIVFPQ.cpp:1033): stop after the probe reaching it
        const int per = (a.nprobe + 63) >> 6;
        const int p0 = lane * per;
        uint64_t local = 0;
        for (int i = 0; i < per; i++) { const int p = p0 + i; if (p < a.nprobe) local += plen[p]; }
        uint64_t incl = local;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t lo = __shfl_up((uint32_t)incl, sft, 64);
            const uint32_t hi = __shfl_up((uint32_t)(incl >> 32), sft, 64);
            const uint64_t o = ((uint64_t)hi << 32) | lo;
            if (lane >= sft) incl += o;
        }
        uint64_t run = incl - local;
        int cut = a.nprobe;                                  // first probe index AFTER the cut
        for (int i = 0; i < per; i++) {
            const int p = p0 + i;
            if (p < a.nprobe) {
                cum[p] = (uint32_t)run;
                run += plen[p];
                if (a.max_codes && run >= (uint64_t)a.max_codes && cut == a.nprobe) cut = p + 1;
            }
        }
        // earliest cut over lanes
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) cut = min(cut, __shfl_xor(cut, sft, 64));
        if (lane == 63) cum[a.nprobe] = (uint32_t)incl;
        if (lane == 0) misc[0] = cut;
    }
    __syncthreads();
    const int np_eff = misc[0];
    if (np_eff < a.nprobe) {
        // probes behind the cut are not visited: make them empty for the emit search
        __syncthreads();
        const uint32_t endpos = cum[np_eff];
        for (int p = np_eff + t; p <= a.nprobe; p += 256) cum[p] = endpos;
        for (int p = np_eff + t; p < a.nprobe; p += 256) pkey[p] = -1;
        __syncthreads();
    }

    WaveSelect<KPL> sel;
    sel.init(a.k, queue + wave * 64, lane);

    // ---- probe loop, software-pipelined one live probe ahead ----------------------
    float4 t2r[4];
    uint4 c0 = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](int p) {
        // first live probe at or after p; returns its index (or np_eff)
        while (p < np_eff && pkey[p] < 0) p++;
        if (p < np_eff) {
            const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)pkey[p] * E);
#pragma unroll
            for (int i = 0; i < 4; i++) t2r[i] = src[i * 256 + t];
            if ((uint32_t)t < plen[p])
                c0 = reinterpret_cast<const uint4*>(a.codes)[poff[p] + t];
        }
        return p;
    };
    int ik = prefetch(0);
    int buf = 0;
    uint64_t nscan = 0;
    while (ik < np_eff) {
        const uint32_t len = plen[ik];
        const float dis0 = pd0[ik];
        const uint32_t pos0 = cum[ik];
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + poff[ik];
        float* L = lut + buf * E;
        // sim_table = term2[key] + (-2) * sim_table_2   (fvec_madd, IndexIVFPQ.cpp:641-644)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float4 s;
            s.x = __fadd_rn(t2r[i].x, m2t3[i].x);
            s.y = __fadd_rn(t2r[i].y, m2t3[i].y);
            s.z = __fadd_rn(t2r[i].z, m2t3[i].z);
            s.w = __fadd_rn(t2r[i].w, m2t3[i].w);
            reinterpret_cast<float4*>(L)[i * 256 + t] = s;
        }
        uint4 cc = c0;
        const int nxt = prefetch(ik + 1);
        __syncthreads();
        for (uint32_t j0 = (uint32_t)wave * 64; j0 < len; j0 += 256) {
            const uint32_t j = j0 + lane;
            const uint32_t jn = j + 256;
            uint4 cn = make_uint4(0, 0, 0, 0);
            if (jn < len) cn = cp[jn];
            const bool valid = j < len;
            float dis = dis0;
            // dis = dis0 + tab[0][c0] + ... + tab[15][c15], left to right (:788-794)
            dis = __fadd_rn(dis, L[0 * 256 + (cc.x & 255u)]);
            dis = __fadd_rn(dis, L[1 * 256 + ((cc.x >> 8) & 255u)]);
            dis = __fadd_rn(dis, L[2 * 256 + ((cc.x >> 16) & 255u)]);
            dis = __fadd_rn(dis, L[3 * 256 + (cc.x >> 24)]);
            dis = __fadd_rn(dis, L[4 * 256 + (cc.y & 255u)]);
            dis = __fadd_rn(dis, L[5 * 256 + ((cc.y >> 8) & 255u)]);
            dis = __fadd_rn(dis, L[6 * 256 + ((cc.y >> 16) & 255u)]);
            dis = __fadd_rn(dis, L[7 * 256 + (cc.y >> 24)]);
            dis = __fadd_rn(dis, L[8 * 256 + (cc.z & 255u)]);
            dis = __fadd_rn(dis, L[9 * 256 + ((cc.z >> 8) & 255u)]);
            dis = __fadd_rn(dis, L[10 * 256 + ((cc.z >> 16) & 255u)]);
            dis = __fadd_rn(dis, L[11 * 256 + (cc.z >> 24)]);
            dis = __fadd_rn(dis, L[12 * 256 + (cc.w & 255u)]);
            dis = __fadd_rn(dis, L[13 * 256 + ((cc.w >> 8) & 255u)]);
            dis = __fadd_rn(dis, L[14 * 256 + ((cc.w >> 16) & 255u)]);
            dis = __fadd_rn(dis, L[15 * 256 + (cc.w >> 24)]);
#ifdef VLQ_EXP_NOSELECT
            sel.offer(dis, pos0 + j, valid && dis < -1e30f);
#else
            sel.offer(dis, pos0 + j, valid);
#endif
            cc = cn;
        }
        nscan += len;
        buf ^= 1;
        ik = nxt;
    }

    merge_and_emit<KPL>(sel, smraw, cum, a, q, wave, lane,
                        [&](int p, int64_t& lkey, int64_t& loff) { lkey = kq[p]; loff = poff[p]; });
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
    if (badkey) *a.bad_key = 1;
}

template <int KPL>
static void launch_scan16_t(const ScanArgs& a, int lut_region, size_t smem, hipStream_t s) {
    static size_t attr_smem = 0;
    if (smem > attr_smem) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan16_kernel<KPL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_smem = smem;
    }
    const unsigned grid = (unsigned)(8 * a.xcd_chunk);
    hipLaunchKernelGGL((scan16_kernel<KPL>), dim3(grid), dim3(256), smem, s, a, lut_region);
}

void launch_scan16(const ScanArgs& a_in, hipStream_t s) {
    if (a_in.nq <= 0) return;
    ScanArgs a = a_in;
    a.xcd_chunk = (int)((a.nq + 7) / 8);
    size_t lutb = (size_t)2 * 4096 * 4;
    const size_t merge = (size_t)4 * a.k * 8;
    if (lutb < merge) lutb = merge;
    const size_t tail = 4 * 64 * 8 + (size_t)a.nprobe * (8 + 4 + 4 + 4 + 4) + 4 + 64;
    const size_t smem = lutb + tail;
    if (a.k <= 64) launch_scan16_t<1>(a, (int)lutb, smem, s);
    else if (a.k <= 256) launch_scan16_t<4>(a, (int)lutb, smem, s);
    else launch_scan16_t<16>(a, (int)lutb, smem, s);
}

// ---------------------------------------------------------------------------
// query order: counting sort of the queries by their nearest coarse centroid
// (keys[q][0]).  Order inside a bin is arbitrary -- it only influences which
// workgroups run next to each other, never a result.
// ---------------------------------------------------------------------------
__global__ void qorder_hist_kernel(const int64_t* __restrict__ keys, int64_t nq, int nprobe,
                                   int nlist, int* __restrict__ hist) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int64_t k0 = keys[q * nprobe];
    atomicAdd(&hist[(k0 >= 0 && k0 < nlist) ? (int)k0 : nlist], 1);
}

__global__ __launch_bounds__(1024) void qorder_scan_kernel(int* __restrict__ hist, int nbins) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nbins + 1023) / 1024;
    const int b0 = t * per;
    int sum = 0;
    for (int i = 0; i < per; i++) if (b0 + i < nbins) sum += hist[b0 + i];
    part[t] = sum;
    __syncthreads();
    for (int sft = 1; sft < 1024; sft <<= 1) {
        const int v = t >= sft ? part[t - sft] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = 0; i < per; i++)
        if (b0 + i < nbins) { const int c = hist[b0 + i]; hist[b0 + i] = run; run += c; }
}

__global__ void qorder_scatter_kernel(const int64_t* __restrict__ keys, int64_t nq, int nprobe,
                                      int nlist, int* __restrict__ hist, int* __restrict__ qorder) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int64_t k0 = keys[q * nprobe];
    const int pos = atomicAdd(&hist[(k0 >= 0 && k0 < nlist) ? (int)k0 : nlist], 1);
    qorder[pos] = (int)q;
}

void launch_query_order(const int64_t* keys, int64_t nq, int nprobe, int nlist, int* hist,
                        int* qorder, hipStream_t s) {
    if (nq <= 0) return;
    (void)hipMemsetAsync(hist, 0, ((size_t)nlist + 1) * sizeof(int), s);
    const unsigned g = (unsigned)((nq + 255) / 256);
    hipLaunchKernelGGL(qorder_hist_kernel, dim3(g), dim3(256), 0, s, keys, nq, nprobe, nlist, hist);
    hipLaunchKernelGGL(qorder_scan_kernel, dim3(1), dim3(1024), 0, s, hist, nlist + 1);
    hipLaunchKernelGGL(qorder_scatter_kernel, dim3(g), dim3(256), 0, s, keys, nq, nprobe, nlist, hist,
                       qorder);
}

}  // namespace vlq
