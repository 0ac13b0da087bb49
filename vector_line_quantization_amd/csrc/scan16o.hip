// List-owned schedule of the 16-byte scan, second build (round 4): the same partitioning as scan16.hip's OWNED mode
// (lists cut into 8 partitions of neighbouring lists, one per XCD; a workgroup = one (query, partition) ITEM = the query's
// probes that fall into that partition, walked in coarse-distance order; raw (distance, scan position) keys out, joined by
// owned_merge_kernel) -- identical arithmetic (IndexIVFPQ.cpp:631-690, :781-802), identical results -- but organised so that
// an item, which holds only ~6 probes of ~330 codes on data with the recall of real descriptors, starts scanning after ONE
// dependent load instead of four:
//   * owned_prep_kernel (one wave per query) does everything that depends on the query alone, once: list offsets and
//     lengths, the scan-position prefix sums and the max_codes cut (what probe_meta_fill / probe_meta_scan redo in every
//     workgroup of the first build -- 5.2 times per query), the split of the probes by partition.  It leaves one 24-byte
//     record per probe, grouped by partition, and per (query, partition) the segment of records that is the item;
//   * the item list of a partition (owned_place2_kernel) carries (query, first record, count) in one 8-byte entry:
//     entry -> records + per-query table row + first term2 row + first codes are three loads deep, the first two of them
//     issued together;
//   * the list loop is the short one (lists of a few hundred codes: no pair loop), which leaves the registers for five
//     workgroups per CU instead of four.
#include <type_traits>

#include "kernels.h"
#include "scan_common.cuh"
#include "scan16_common.cuh"
#include "wave_topk.cuh"

namespace vlq {

// ---------------------------------------------------------------------------
// per-query preparation: one wave per query, lane = probe (nprobe <= 64)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void owned_prep_kernel(const int64_t* __restrict__ keys, const float* __restrict__ coarse_dis,
                                                         int64_t nq, int nprobe, int nlist, const int64_t* __restrict__ list_off,
                                                         const int64_t* __restrict__ list_len, int64_t max_codes,
                                                         const int* __restrict__ list_rank, const uint8_t* __restrict__ list_part,
                                                         OwnRec* __restrict__ recs, uint32_t* __restrict__ seg, int* __restrict__ minr,
                                                         int* __restrict__ hist, uint8_t* __restrict__ part_mask, int* __restrict__ bad_key) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const bool valid = lane < nprobe;
    const int64_t key = valid ? keys[q * nprobe + lane] : -1;
    if (key >= nlist) *bad_key = 1;                              // IndexIVFPQ.cpp:1008-1011
    const bool live = key >= 0 && key < nlist;
    int64_t off = 0, len = 0;
    if (live) { off = list_off[key]; len = list_len ? list_len[key] : list_off[key + 1] - off; }
    // scan positions: exclusive prefix of the lengths in coarse order; the max_codes cut (IndexIVFPQ.cpp:1033: stop after
    // the probe that reaches it) exactly as probe_meta_scan applies it
    uint64_t incl = (uint64_t)len;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)incl, sft, 64), hi = __shfl_up((uint32_t)(incl >> 32), sft, 64);
        if (lane >= sft) incl += ((uint64_t)hi << 32) | lo;
    }
    int cut = (valid && max_codes && incl >= (uint64_t)max_codes) ? lane + 1 : nprobe;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) cut = min(cut, __shfl_xor(cut, sft, 64));
    const bool vis = live && len > 0 && lane < cut;              // empty lists are skipped (:1016)
    const int part = vis ? (int)list_part[key] : -1;
    const int rank = vis ? list_rank[key] : 0x7fffffff;
    const u64 lt = (1ull << lane) - 1ull;
    int start = 0, my_slot = 0;
    uint32_t mask = 0, my_seg = 0;
    int my_minr = 0x7fffffff;
#pragma unroll
    for (int x = 0; x < 8; x++) {
        const u64 m = __ballot(part == x);
        const int n = __popcll(m);
        int r = part == x ? rank : 0x7fffffff;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) r = min(r, __shfl_xor(r, sft, 64));
        if (part == x) my_slot = start + __popcll(m & lt);
        if (lane == x) { my_seg = (uint32_t)start | ((uint32_t)n << 16); my_minr = r; }
        if (n) mask |= 1u << x;
        start += n;
    }
    if (vis) {
        OwnRec rc;
        rc.key = (int32_t)key; rc.len = (uint32_t)len; rc.off = off; rc.dis0 = coarse_dis[q * nprobe + lane];
        rc.pos0 = (uint32_t)(incl - (uint64_t)len);
        recs[q * nprobe + my_slot] = rc;
    }
    if (lane < 8) {
        seg[q * 8 + lane] = my_seg;
        minr[q * 8 + lane] = my_minr;
        if (my_minr != 0x7fffffff) atomicAdd(&hist[(size_t)lane * nlist + my_minr], 1);
    }
    if (lane == 0) part_mask[q] = (uint8_t)mask;
}

// item list of every partition: the queries with probes there, in the order of the spatial rank of their nearest owned list
// (counting sort: hist = bin counts from the prep kernel); an entry = (query, segment of its records)
__global__ __launch_bounds__(256) void owned_place2_kernel(int64_t nq, int nlist, const int* __restrict__ hist, int* __restrict__ cnt,
                                                           const int* __restrict__ minr, const uint32_t* __restrict__ seg,
                                                           uint2* __restrict__ items, int* __restrict__ own_count) {
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
    items[(int64_t)x * nq + pre[r] + atomicAdd(&cnt[(size_t)x * nlist + r], 1)] = make_uint2((uint32_t)q, seg[q * 8 + x]);
}

void launch_owned2_prepare(const ScanArgs& a, const int* list_rank, int* hist, int* minr, uint32_t* seg, uint2* items,
                           int* own_count, uint8_t* part_mask, OwnRec* recs, hipStream_t s) {
    if (a.nq <= 0) return;
    (void)hipMemsetAsync(hist, 0, (size_t)16 * a.nlist * sizeof(int), s);     // hist | cnt
    hipLaunchKernelGGL(owned_prep_kernel, dim3((unsigned)((a.nq + 3) / 4)), dim3(256), 0, s, a.keys, a.coarse_dis, a.nq, a.nprobe,
                       a.nlist, a.list_off, a.list_len, a.max_codes, list_rank, a.list_part, recs, seg, minr, hist, part_mask,
                       a.bad_key);
    const size_t smem = (size_t)a.nlist * sizeof(int);
    ensure_dynamic_lds(reinterpret_cast<const void*>(owned_place2_kernel), smem);
    hipLaunchKernelGGL(owned_place2_kernel, dim3((unsigned)((a.nq + 255) / 256), 8), dim3(256), smem, s, a.nq, a.nlist, hist,
                       hist + (size_t)8 * a.nlist, minr, seg, items, own_count);
}

// ---------------------------------------------------------------------------
// the scan of one item
// ---------------------------------------------------------------------------
template <int KPL, int NBUF>
__global__ __launch_bounds__(256) void scan16o_kernel(ScanArgs a, int lut_region) {
    constexpr int E = 4096, NW = 4, NT = 256, NI = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* lut = reinterpret_cast<float*>(smraw);                         // [NBUF][E] at LDS byte 0
    u64* queue = reinterpret_cast<u64*>(smraw + lut_region);              // [NW][64]
    uint32_t* recl = reinterpret_cast<uint32_t*>(queue + NW * 64);        // [nprobe][6] this item's records
    uint32_t* wg_thr = recl + a.nprobe * 6;                               // min of the waves' k-th distances

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef VLQ_PHASE_TIMING
    const uint64_t tk0 = wall_clock64();
#endif
    if (__builtin_amdgcn_groupstaticsize() != 0) { *a.bad_key = 2; return; }   // adc16_halves addresses the buffers at 0 / 16384
    uint32_t two = 2;
    asm volatile("" : "+v"(two));
    // Consecutive workgroups go round-robin over the 8 XCDs.  XCD c takes the c-th EIGHTH of the items in partition-major
    // order: partition c's items, give or take the ends it shares with its spatial neighbours -- the partitions hold equal
    // numbers of lists, not of items (measured on the bench data: 6599 ... 7969 items, the fullest XCD 14.5 % over the mean).
    int x = 0;
    int64_t slot = 0;
    {
        const int c = (int)(blockIdx.x & 7);
        int cnt[8], total = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { cnt[i] = a.own_count[i]; total += cnt[i]; }
        const int per = (total + 7) >> 3;
        const int64_t g = (int64_t)c * per + (int64_t)(blockIdx.x >> 3);
        if ((int64_t)(blockIdx.x >> 3) >= per || g >= total) return;
        int64_t base = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) if (g >= base + cnt[i] && x == i) { base += cnt[i]; x = i + 1; }
        slot = g - base;
    }
    const uint2 ent = a.own_items[(int64_t)x * a.nq + slot];
    const int64_t q = ent.x;
    const int start = (int)(ent.y & 0xffffu), n = (int)(ent.y >> 16);

    // the item's records into LDS, the per-query table (-2 <q_m, cent_mj>, materialised by qtab16_kernel) into registers
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.own_recs + q * a.nprobe + start);
        if (t < n * 6) recl[t] = src[t];
        if (t == 0) *wg_thr = f32_to_ordered(3.402823466e+38f);
    }
    float4 m2t3[NI];
    {
        const float4* qt = reinterpret_cast<const float4*>(a.qtab + q * E);
#pragma unroll
        for (int i = 0; i < NI; i++) m2t3[i] = qt[i * NT + t];
    }
    WaveSelect<KPL, 1, (KPL >= 2)> sel;
    sel.init(a.k, queue + wave * 64, lane);
    __syncthreads();

    // rows and first codes are requested one probe ahead.  (Two probes ahead -- a second register set, 101 VGPRs = 4
    // workgroups per CU instead of 5 -- shortens an item's loop from 12.5 to 11.2 us and loses the occupancy: 0.87 against
    // 0.83 ms for the batch.)
    float4 ra[NI];
    uint4 ca = make_uint4(0, 0, 0, 0);
    auto fetch = [&](int i, float4 (&r)[NI], uint4& c) __attribute__((always_inline)) {
        if (i >= n) return;
        const uint32_t* rec = recl + i * 6;                      // {key, len, off lo, off hi, dis0, pos0}
        const int key = __builtin_amdgcn_readfirstlane(rec[0]);
        const uint32_t len = __builtin_amdgcn_readfirstlane(rec[1]);
        const int64_t off = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rec[3]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rec[2]));
        const float4* src = reinterpret_cast<const float4*>(a.term2 + (size_t)key * E);
#pragma unroll
        for (int i2 = 0; i2 < NI; i2++) r[i2] = src[i2 * NT + t];
        c = (reinterpret_cast<const uint4*>(a.codes) + off)[min((uint32_t)t, len - 1)];
    };
    fetch(0, ra, ca);
#ifdef VLQ_PHASE_TIMING
    const uint64_t tk1 = wall_clock64();
#endif
    int buf = 0;
    uint32_t nscan = 0;
    auto probe = [&](int i, float4 (&r)[NI], uint4& c) __attribute__((always_inline)) {
        const uint32_t* rec = recl + i * 6;
        const uint32_t len = __builtin_amdgcn_readfirstlane(rec[1]);
        const int64_t off = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rec[3]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rec[2]));
        const float dis0 = __uint_as_float(__builtin_amdgcn_readfirstlane(rec[4]));
        const uint32_t pos0 = __builtin_amdgcn_readfirstlane(rec[5]);
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + off;
        float* L = lut + buf * E;
        if (NBUF == 1) __syncthreads();                          // everyone is done scanning with the single buffer
        __builtin_amdgcn_s_setprio(2);                           // table build + next loads first (scan16.hip)
        build_lut16<NI>(L, t, r, m2t3);
        uint4 cc = c;
        fetch(i + 1, r, c);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (sel.dirty) {                                         // publish this wave's k-th distance, take the workgroup's minimum
            if (lane == 0) atomicMin(wg_thr, f32_to_ordered(sel.thr_own));
            sel.dirty = false;
        }
        sel.refresh_with(*wg_thr);
        auto scan_list = [&](auto bufc) {
            constexpr int B = decltype(bufc)::value;
            for (uint32_t j0 = (uint32_t)wave * 64; j0 < len; j0 += NT) {
                const uint32_t j = j0 + lane;
                const float dis = adc16_halves<B>(cc, dis0, two);
                sel.offer_keyed(dis, pos0 + j, j < len);
                if (j0 + NT < len) cc = cp[min(j + NT, len - 1)];   // (wave-uniform) most lists end within the trip
            }
        };
        if (NBUF == 1 || buf == 0) scan_list(std::integral_constant<int, 0>{});
        else scan_list(std::integral_constant<int, 1>{});
        nscan += len;
        if (NBUF == 2) buf ^= 1;
    };
    for (int i = 0; i < n; i++) probe(i, ra, ca);
#ifdef VLQ_PHASE_TIMING
    const uint64_t tk2 = wall_clock64();
#endif
    // raw keys out: scan positions are global to the query, so owned_merge_kernel orders the parts' candidates exactly
    // like one workgroup scanning all probes would have
    if (KPL == 1 && a.k <= 16) {
        // the four waves' k best are at most 64 keys: one 64-key sort instead of three merges (3.0 -> ~0.5 us per item)
        sel.flush();
        __syncthreads();                                         // the table is free
        u64* mb = reinterpret_cast<u64*>(smraw);
        if (lane < 16) mb[wave * 16 + lane] = lane < a.k ? sel.best[0] : kMaxKey;
        __syncthreads();
        if (wave == 0) {
            const u64 key = wave_sort64(mb[lane], lane);
            if (lane < a.k) a.part_keys[((size_t)q * 8 + x) * a.k + lane] = key;
        }
    } else if (merge_waves<KPL, NW>(sel, smraw, a.k, wave, lane)) {
        u64* out = a.part_keys + ((size_t)q * 8 + x) * a.k;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const int e = r * 64 + lane;
            if (e < a.k) out[e] = sel.best[r];
        }
    }
    if (t == 0) atomicAdd(a.ncode, (unsigned long long)nscan);
#ifdef VLQ_PHASE_TIMING
    if (t == 0 && (blockIdx.x % 61) == 0) {      // a sample: the four atomics of every workgroup would be what is measured
        const uint64_t tk3 = wall_clock64();
        atomicAdd(a.ncode + 2, (unsigned long long)(tk1 - tk0));
        atomicAdd(a.ncode + 3, (unsigned long long)(tk2 - tk1));
        atomicAdd(a.ncode + 4, (unsigned long long)(tk3 - tk2));
        atomicAdd(a.ncode + 5, 1ull);
    }
#endif
}

template <int KPL, int NBUF>
static void launch_scan16o_t(const ScanArgs& a, hipStream_t s) {
    size_t lutb = (size_t)NBUF * 4096 * 4;
    const size_t merge = (size_t)4 * a.k * 8;
    if (lutb < merge) lutb = merge;
    const size_t smem = lutb + 4 * 64 * 8 + (size_t)a.nprobe * 24 + 16;
    ensure_dynamic_lds(reinterpret_cast<const void*>(scan16o_kernel<KPL, NBUF>), smem);
    hipLaunchKernelGGL((scan16o_kernel<KPL, NBUF>), dim3((unsigned)(8 * a.nq)), dim3(256), smem, s, a, (int)lutb);
}

bool scan16o_supports(const ScanArgs& a) { return a.nprobe <= 64 && a.k <= 256 && a.M == 16 && a.ksub == 256; }

void launch_scan16_owned2(const ScanArgs& a, int nbuf, hipStream_t s) {
    if (a.nq <= 0) return;
    if (nbuf == 2) {
        if (a.k <= 64) launch_scan16o_t<1, 2>(a, s);
        else if (a.k <= 128) launch_scan16o_t<2, 2>(a, s);
        else launch_scan16o_t<4, 2>(a, s);
    } else {
        if (a.k <= 64) launch_scan16o_t<1, 1>(a, s);
        else if (a.k <= 128) launch_scan16o_t<2, 1>(a, s);
        else launch_scan16o_t<4, 1>(a, s);
    }
}

}  // namespace vlq
