// C ABI (include/vlq_ivfpq.h) over the HIP kernels.  Host-side orchestration only:
// device buffers, workspace, paging of large query batches, host<->device staging.
// No CPU compute path exists here: every search/add entry point launches kernels.
#include "handle.h"
#include "line.h"
#include "lists.h"

namespace vlq_detail {

int set_dev(vlq_ivfpq_t h) {
    HIP_TRY(hipSetDevice(h->device));
    return VLQ_OK;
}

// stage an input: returns a device pointer holding `bytes` of src
int stage_in(vlq_ivfpq_t h, const void* src, size_t bytes, DevBuf& ws, const void** out) {
    if (bytes == 0) { *out = src; return VLQ_OK; }
    if (is_device_ptr(src)) { *out = src; return VLQ_OK; }
    TRY(ws.reserve(bytes));
    HIP_TRY(hipMemcpyAsync(ws.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    *out = ws.p;
    return VLQ_OK;
}

// pick the device-side destination of an output.  zero_copy != nullptr: page-locked host memory is written by the
// kernels themselves (the rows cross PCIe while the scan runs; the caller synchronises the stream before returning)
int stage_out(void* dst, size_t bytes, DevBuf& ws, void** dev, bool* need_copy, bool* zero_copy) {
    if (zero_copy) *zero_copy = false;
    void* mapped = nullptr;
    const int kind = ptr_kind(dst, &mapped);
    if (kind == 1) { *dev = dst; *need_copy = false; return VLQ_OK; }
    if (kind == 2 && zero_copy) { *dev = mapped; *need_copy = false; *zero_copy = true; return VLQ_OK; }
    TRY(ws.reserve(bytes));
    *dev = ws.p;
    *need_copy = true;
    return VLQ_OK;
}

hipEvent_t get_event(vlq_ivfpq_t h) {
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct StageTimer {
    vlq_ivfpq_t h; int stage; hipEvent_t a = nullptr, b = nullptr;
    StageTimer(vlq_ivfpq_t h_, int stage_) : h(h_), stage(stage_) {
        if (!h->prof || (h->prof_scan_only && stage != 2)) return;
        if (h->prof_every > 1 && stage == 2 && (h->prof_seq++ % (uint64_t)h->prof_every) != 0) return;
        a = get_event(h); b = get_event(h);
        if (a) (void)hipEventRecord(a, h->stream);
    }
    void stop() {
        if (!h->prof || !a || !b) return;
        (void)hipEventRecord(b, h->stream);
        h->pending.push_back({a, b, stage});
        a = b = nullptr;
    }
};

void drain_profile(vlq_ivfpq_t h) {
    for (auto& p : h->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->prof_ms[p.stage] += ms;
            if (p.stage == 2) h->prof_calls++;
        }
        h->ev_pool.push_back(p.a);
        h->ev_pool.push_back(p.b);
    }
    h->pending.clear();
}

int ensure_term2(vlq_ivfpq_t h) {
    if (!(h->by_residual && h->use_precomputed_table == 1)) return VLQ_OK;
    if (h->term2_valid) return VLQ_OK;
    if (!h->have_coarse || !h->have_pq) return fail(VLQ_ERR_STATE, "centroids not set");
    const size_t E = (size_t)h->M * h->ksub;
    if (h->imi_nbits > 0) {
        // table type 2 (IndexIVFPQ.cpp:430-457): one row per coarse sub-centroid index
        const int64_t kc = int64_t(1) << h->imi_nbits;
        TRY(h->term2.reserve((size_t)kc * E * sizeof(float)));
        vlq::launch_pq_tables(h->imi_virtual.as<float>(), kc, h->d, h->pq.as<float>(), h->M, h->ksub,
                              h->dsub, h->rnorm.as<float>(), 2, h->term2.as<float>(), h->stream);
        HIP_TRY(hipGetLastError());
        h->term2_valid = true;
        return VLQ_OK;
    }
    TRY(h->term2.reserve((size_t)h->nlist * E * sizeof(float)));
    // IndexIVFPQ::precompute_table (IndexIVFPQ.cpp:411-429)
    vlq::launch_pq_tables(h->coarse.as<float>(), h->nlist, h->d, h->pq.as<float>(), h->M, h->ksub,
                          h->dsub, h->rnorm.as<float>(), 2, h->term2.as<float>(), h->stream);
    HIP_TRY(hipGetLastError());
    h->term2_valid = true;
    return VLQ_OK;
}

// half(term 2) for the float16 tables (impl/IVFPQ.cu:599-684 toHalf).  As in the reference the entries must fit
// the half range: byte-valued (SIFT-like) data has |term 2| up to 1e5 and would turn into infinities -- refused.
int ensure_term2h(vlq_ivfpq_t h) {
    TRY(ensure_term2(h));
    if (h->term2h_valid) return VLQ_OK;
    const int64_t n = (int64_t)h->nlist * h->M * h->ksub;
    TRY(h->ws_misc.reserve(16));
    vlq::launch_max_abs(h->term2.as<float>(), n, h->ws_misc.as<unsigned int>(), h->stream);
    unsigned int mx = 0;
    HIP_TRY(hipMemcpyAsync(&mx, h->ws_misc.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float mxf;
    memcpy(&mxf, &mx, 4);
    if (!(mxf <= 65504.f))
        return fail(VLQ_ERR_UNSUPPORTED, "float16 look-up tables: |term 2| reaches %g, beyond the half range (65504); "
                    "use fp32 tables for this data (the reference's half tables would hold infinities)", (double)mxf);
    TRY(h->term2h.reserve((size_t)n * 2));
    vlq::launch_to_half(h->term2.as<float>(), n, 1.f, h->term2h.as<uint16_t>(), h->stream);
    HIP_TRY(hipGetLastError());
    h->term2h_valid = true;
    return VLQ_OK;
}

int check_ready(vlq_ivfpq_t h, bool need_lists) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (!h->have_coarse) return fail(VLQ_ERR_STATE, "coarse centroids not set (index not trained)");
    if (!h->have_pq) return fail(VLQ_ERR_STATE, "PQ centroids not set (index not trained)");
    if (need_lists && !h->have_lists) return fail(VLQ_ERR_STATE, "inverted lists not loaded");
    return VLQ_OK;
}

int64_t query_page(vlq_ivfpq_t h) {
    // GpuIndex::search pages at 32768 queries (gpu/GpuIndex.cu:29,108-147); also keep the
    // [page][nlist] distance matrix under 8 GiB (sized for 288 GB of HBM: at 2^17 lists a 10 000-query
    // batch is one 5.2 GB page; 1 GiB pages cost the coarse stage 15 % there)
    int64_t page = 32768;
    int64_t by_mat = (int64_t)((size_t(1) << 31) / (size_t)std::max(1, h->nlist));
    page = std::max<int64_t>(1, std::min(page, by_mat));
    return page;
}

// the screen's "rows it could not decide" counter: on the device, mirrored into page-locked host memory behind every batch
static int screen_counters(vlq_ivfpq_t h) {
    if (h->screen_cnt_host) return VLQ_OK;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->screen_cnt_host), 8, hipHostMallocDefault));
    *h->screen_cnt_host = 0;
    TRY(h->ws_screen_cnt.reserve(8));
    HIP_TRY(hipMemsetAsync(h->ws_screen_cnt.p, 0, 8, h->stream));
    return VLQ_OK;
}
static int screen_counters_copy(vlq_ivfpq_t h, int64_t n) {
    h->screen_rows_seen += (uint64_t)n;
    HIP_TRY(hipMemcpyAsync(h->screen_cnt_host, h->ws_screen_cnt.p, 4, hipMemcpyDeviceToHost, h->stream));
    h->screen_rows_copied = h->screen_rows_seen;
    return VLQ_OK;
}

// coarse stage of one page; keep_matrix: the caller reads the [n][nlist] distance matrix in h->ws_dist
// afterwards (VLQ line select)
int coarse_page(vlq_ivfpq_t h, int64_t n, const float* x_dev, int nprobe, float* cdis_dev,
                int64_t* keys_dev, bool zero_qnorm, bool direct, bool keep_matrix) {
    TRY(h->ws_qn.reserve((size_t)n * sizeof(float)));
    // 1-NN (assignment): per-tile (distance, column) keys instead of the [n][nlist] matrix
    const bool argmin = nprobe == 1 && !direct && !keep_matrix && vlq::coarse_argmin_ok(h->nlist, h->d);
    float* tmin = nullptr;
    int fs = 0, fcap = 0;
    const bool filtered = !direct && !keep_matrix && !argmin && !zero_qnorm && h->coarse_filter &&
                          vlq::coarse_filter_ok(h->nlist, h->d, nprobe, n, &fs, &fcap);
    const int64_t n_pad = (n + 127) / 128 * 128;       // whole 128-row blocks: the pipelined distance kernel stores without a row guard
    if (!argmin && !filtered) TRY(h->ws_dist.reserve((size_t)n_pad * h->nlist * sizeof(float)));
    if (filtered) {
        // filtered coarse stage: no [n][nlist] matrix.  (1) exact distances to a sample of the column tiles
        // and their nprobe smallest -> the nprobe-th is an upper bound of the row's nprobe-th smallest overall;
        // (2) the full pass keeps only elements at or below the bound; (3) exact select over the kept keys.
        const int ns = h->nlist / fs;
        if (h->coarse_s_stride != fs) {
            TRY(h->coarse_s.reserve((size_t)ns * h->d * sizeof(float)));
            TRY(h->cnorm_s.reserve((size_t)ns * sizeof(float)));
            vlq::launch_sample_tiles(h->coarse.as<float>(), h->cnorm.as<float>(), h->nlist, h->d, fs, h->coarse_s.as<float>(),
                                     h->cnorm_s.as<float>(), h->stream);
            h->coarse_s_stride = fs;
        }
        TRY(h->ws_dist.reserve((size_t)n * ns * sizeof(float)));
        const size_t ntl = (size_t)h->nlist / 64;
        TRY(h->ws_cand.reserve((size_t)n * ntl * fcap * 8));
        TRY(h->ws_cnt.reserve((size_t)n * ntl));
        vlq::launch_row_norms(x_dev, n, h->d, h->ws_qn.as<float>(), h->stream);
        vlq::launch_coarse_distances(x_dev, h->coarse_s.as<float>(), h->ws_qn.as<float>(), h->cnorm_s.as<float>(),
                                     h->ws_dist.as<float>(), n, ns, h->d, h->stream, nullptr);
        vlq::launch_coarse_select(h->ws_dist.as<float>(), n, ns, nprobe, cdis_dev, keys_dev, h->stream, nullptr);
        vlq::launch_coarse_distances_filtered(x_dev, h->coarse.as<float>(), h->ws_qn.as<float>(), h->cnorm.as<float>(), n,
                                              h->nlist, h->d, cdis_dev + (nprobe - 1), nprobe,
                                              h->ws_cand.as<unsigned long long>(), h->ws_cnt.as<unsigned char>(), h->stream);
        vlq::launch_coarse_select_cand(h->ws_cand.as<unsigned long long>(), h->ws_cnt.as<unsigned char>(), n, nprobe, cdis_dev,
                                       keys_dev, x_dev, h->coarse.as<float>(), h->ws_qn.as<float>(), h->cnorm.as<float>(),
                                       h->nlist, h->d, h->stream);
        HIP_TRY(hipGetLastError());
        return VLQ_OK;
    }
    if (h->coarse_screen && h->screen_cnt_host && h->screen_rows_copied >= 1024 &&
        (uint64_t)*h->screen_cnt_host * 200 > h->screen_rows_copied)
        h->coarse_screen = 0;                     // this index's data defeat the screen's bound: matrix path from here on
    if (argmin && !zero_qnorm && h->coarse_screen && h->screen.ok && n >= 2048 && vlq::coarse_screen_nn_shape_ok(h->nlist, h->d)) {
        // 1-NN (add / encode): approximate tile minima only, the tiles under the bound exactly (coarse_screen.hip)
        TRY(screen_counters(h));
        const int dp = (h->d + 15) / 16 * 16;
        TRY(h->ws_xh.reserve((size_t)n_pad * dp * 2));
        TRY(h->ws_xflags.reserve((size_t)n));
        TRY(h->ws_qn_c.reserve((size_t)n * sizeof(float)));
        TRY(h->ws_tmin.reserve((size_t)n * (h->nlist / 64) * 8));
        vlq::launch_screen_prep(x_dev, h->screen.mu.as<float>(), n, h->d, h->screen.scale, h->ws_xh.p, h->ws_qn.as<float>(),
                                h->ws_qn_c.as<float>(), h->ws_xflags.as<unsigned char>(), h->stream);
        vlq::launch_coarse_screened_nn(x_dev, h->ws_xh.p, h->ws_xflags.as<unsigned char>(), h->coarse.as<float>(), h->screen.half.p,
                                       h->ws_qn.as<float>(), h->cnorm.as<float>(), h->ws_qn_c.as<float>(), h->screen.norm_c.as<float>(),
                                       h->ws_tmin.as<float>(), n, h->nlist, h->d, h->screen.scale, h->screen.cmax, h->screen.cmax0, cdis_dev,
                                       keys_dev, h->ws_screen_cnt.as<unsigned int>(), h->stream);
        TRY(screen_counters_copy(h, n));
        HIP_TRY(hipGetLastError());
        return VLQ_OK;
    }
    // (below ~2000 rows the screen's five short kernels cost more than the matrix path's two: 1250 rows 46 against 40 us)
    if (!direct && !keep_matrix && !argmin && !zero_qnorm && h->coarse_screen && h->screen.ok && n >= 2048 &&
        vlq::coarse_screen_shape_ok(h->nlist, h->d, nprobe)) {
        TRY(screen_counters(h));
        // float16 screen (coarse_screen.hip): approximate matrix -> kept columns -> exact fmaf chains -> exact select
        const int dp = (h->d + 15) / 16 * 16;
        TRY(h->ws_xh.reserve((size_t)n_pad * dp * 2));
        TRY(h->ws_xflags.reserve((size_t)n));
        static const bool screen_stats = getenv("VLQ_SCREEN_STATS") != nullptr;       // kept columns per row, printed at destroy
        if (screen_stats && !h->ws_kept.p) {
            TRY(h->ws_kept.reserve(16));
            HIP_TRY(hipMemsetAsync(h->ws_kept.p, 0, 16, h->stream));
        }
        TRY(h->ws_cand.reserve(vlq::coarse_screen_keep_bytes(n, h->nlist)));
        if (h->nlist > 8192 || vlq::coarse_screen_matrix_free_ok(h->nlist, nprobe)) TRY(h->ws_tmin.reserve((size_t)n * (h->nlist / 16 + 32) * sizeof(float)));
        TRY(h->ws_qn_c.reserve((size_t)n * sizeof(float)));
        vlq::launch_screen_prep(x_dev, h->screen.mu.as<float>(), n, h->d, h->screen.scale, h->ws_xh.p, h->ws_qn.as<float>(),
                                h->ws_qn_c.as<float>(), h->ws_xflags.as<unsigned char>(), h->stream);
        vlq::launch_coarse_screened(x_dev, h->ws_xh.p, h->ws_xflags.as<unsigned char>(), h->coarse.as<float>(), h->screen.half.p,
                                    h->ws_qn.as<float>(), h->cnorm.as<float>(), h->ws_qn_c.as<float>(), h->screen.norm_c.as<float>(),
                                    h->ws_dist.as<float>(), h->ws_tmin.p ? h->ws_tmin.as<float>() : nullptr, h->ws_cand.p, n, h->nlist,
                                    h->d, nprobe, h->screen.scale, h->screen.cmax, h->screen.cmax0, cdis_dev, keys_dev,
                                    h->ws_kept.p ? h->ws_kept.as<unsigned long long>() : nullptr, h->ws_screen_cnt.as<unsigned int>(),
                                    h->stream, h->order_hist, &h->order_hist_ready);
        TRY(screen_counters_copy(h, n));
        HIP_TRY(hipGetLastError());
        return VLQ_OK;
    }
    if (direct) {
        vlq::launch_coarse_distances_direct(x_dev, h->coarse.as<float>(), h->ws_dist.as<float>(), n,
                                            h->nlist, h->d, h->stream);
    } else {
        // |q|^2: zeros for the VLQ path; otherwise computed inside the distance kernel (d <= 128) or by its own launch
        const bool fused_norms = !zero_qnorm && vlq::coarse_norms_fused_ok(h->d);
        if (zero_qnorm) HIP_TRY(hipMemsetAsync(h->ws_qn.p, 0, (size_t)n * sizeof(float), h->stream));
        else if (!fused_norms) vlq::launch_row_norms(x_dev, n, h->d, h->ws_qn.as<float>(), h->stream);
        if (argmin) {
            TRY(h->ws_tmin.reserve((size_t)n * (h->nlist / 64) * 8));
            tmin = h->ws_tmin.as<float>();
        } else if (vlq::coarse_tile_minima_ok(h->nlist, h->d, nprobe)) {
            TRY(h->ws_tmin.reserve((size_t)n * (h->nlist / 64) * sizeof(float)));
            tmin = h->ws_tmin.as<float>();
        }
        vlq::launch_coarse_distances(x_dev, h->coarse.as<float>(), fused_norms ? nullptr : h->ws_qn.as<float>(),
                                     h->cnorm.as<float>(), argmin ? nullptr : h->ws_dist.as<float>(), n,
                                     h->nlist, h->d, h->stream, tmin, argmin ? 0 : n_pad);
    }
    if (argmin)
        vlq::launch_coarse_argmin(tmin, n, h->nlist, cdis_dev, keys_dev, h->stream);
    else
        vlq::launch_coarse_select(h->ws_dist.as<float>(), n, h->nlist, nprobe, cdis_dev, keys_dev,
                                  h->stream, tmin);
    HIP_TRY(hipGetLastError());
    return VLQ_OK;
}

// coarse stage on device buffers: x_dev [n][d] -> cdis_dev, keys_dev [n][nprobe]
// MultiIndexQuantizer::search (IndexPQ.cpp:804-857) for one page: the two distance tables,
// their T smallest entries in order, then the MinSumK walk
int imi_page(vlq_ivfpq_t h, int64_t n, const float* x_dev, int k, float* cdis_dev, int64_t* keys_dev) {
    const int kc = 1 << h->imi_nbits, dc = h->d / 2;
    const int T = std::min(k, kc);
    // workspace: 2 tables [n][kc] | sorted values 2x[n][T] | sorted ids 2x[n][T] | heap
    const int64_t n_pad = (n + 127) / 128 * 128;
    const size_t b_tab = (size_t)n_pad * kc * 4, b_sv = (size_t)n * T * 4, b_si = (size_t)n * T * 8;
    // (heap rows in global memory: only the thread-per-query replay of kernels.hip beyond its LDS sizes needs them)
    const bool heap_rows = !(k <= 64 || vlq::imi_minsum_wide_ok(T, k, kc)) || getenv("VLQ_IMI_MINSUM_WIDE_FROM") || getenv("VLQ_IMI_MINSUM_LDS");
    const size_t b_hv = heap_rows ? (size_t)n * 2 * k * 4 : 0, b_hi = heap_rows ? (size_t)n * 2 * k * 8 : 0, b_sub = (size_t)n * dc * 4;
    TRY(h->ws_imi.reserve(2 * b_tab + 2 * b_sv + 2 * b_si + b_hv + b_hi + 2 * b_sub + 256));
    char* p = h->ws_imi.as<char>();
    float* tab[2] = {(float*)p, (float*)(p + b_tab)};
    p += 2 * b_tab;
    int64_t* si[2] = {(int64_t*)p, (int64_t*)(p + b_si)};
    p += 2 * b_si;
    int64_t* hi = (int64_t*)p;
    p += b_hi;
    float* sv[2] = {(float*)p, (float*)(p + b_sv)};
    p += 2 * b_sv;
    float* hv = (float*)p;
    p += b_hv;
    float* sub = (float*)p;
    float* sub2 = (float*)(p + b_sub);
    if (h->coarse_screen && h->screen_cnt_host && h->screen_rows_copied >= 1024 &&
        (uint64_t)*h->screen_cnt_host * 200 > h->screen_rows_copied)
        h->coarse_screen = 0;                     // this index's data defeat the screen's bound: matrix path from here on
    // the radix select + one sort of imi_wide.hip against the running wave selection of kernels.hip (which stops at 1024):
    // coarse stage of 10 000 queries on 2 x 14 bits at 256 / 512 / 1024 cells 2.09 / 4.93 / 10.5 ms with the wave selection,
    // 2.73 / 4.22 / 8.69 with the radix select
    static const int radix_from = [] { const char* e = getenv("VLQ_IMI_RADIX_FROM"); return e ? atoi(e) : 400; }();
    int64_t screened_rows = 0;       // rows that went through the two-pass screen (either half), counted once behind the join
    for (int m = 0; m < 2; m++) {
        const float* cent = h->imi_cent.as<float>() + (size_t)m * kc * dc;
        float* tmin = nullptr;
        bool argmin = false;
        if (dc >= 16 && T == 1 && h->coarse_screen && h->imi_screen[m].ok && n >= 2048 && vlq::coarse_screen_nn_shape_ok(kc, dc)) {
            // the assignment of add / encode: nearest sub-centroid of each half, tile minima only (coarse_screen.hip)
            const vlq_ivfpq_s::ScreenSet& sc = h->imi_screen[m];
            const int dp = (dc + 15) / 16 * 16;
            TRY(screen_counters(h));
            TRY(h->ws_xh.reserve((size_t)n_pad * dp * 2));
            TRY(h->ws_xflags.reserve((size_t)n));
            TRY(h->ws_qn.reserve((size_t)n * 4));
            TRY(h->ws_qn_c.reserve((size_t)n * 4));
            TRY(h->ws_tmin.reserve((size_t)n * (kc / 64) * 8));
            vlq::launch_gather_cols(x_dev, n, h->d, m * dc, dc, sub, h->stream);
            vlq::launch_screen_prep(sub, sc.mu.as<float>(), n, dc, sc.scale, h->ws_xh.p, h->ws_qn.as<float>(), h->ws_qn_c.as<float>(),
                                    h->ws_xflags.as<unsigned char>(), h->stream);
            vlq::launch_coarse_screened_nn(sub, h->ws_xh.p, h->ws_xflags.as<unsigned char>(), cent, sc.half.p, h->ws_qn.as<float>(),
                                           h->imi_norm.as<float>() + (size_t)m * kc, h->ws_qn_c.as<float>(), sc.norm_c.as<float>(),
                                           h->ws_tmin.as<float>(), n, kc, dc, sc.scale, sc.cmax, sc.cmax0, sv[m], si[m],
                                           h->ws_screen_cnt.as<unsigned int>(), h->stream);
            TRY(screen_counters_copy(h, n));
            continue;
        }
        if (dc >= 16 && h->coarse_screen && h->imi_screen[m].ok && n >= 2048 && vlq::coarse_screen_shape_ok(kc, dc, T)) {
            // float16 screen of this half's table (coarse_screen.hip): approximate half matrix in tab[m], kept columns, exact
            // fmaf chains, exact select -- the T nearest sub-centroids and their distances as the matrix path returns them.
            // Round 5: the two halves are independent chains of six latency-bound kernels (~140 us each at 2 x 14 bits); the
            // second runs beside the first on an auxiliary stream with its own per-half workspaces, joined before the MinSumK
            // replay.
            const vlq_ivfpq_s::ScreenSet& sc = h->imi_screen[m];
            const int dp = (dc + 15) / 16 * 16;
            static const bool one_stream = getenv("VLQ_IMI_ONE_STREAM") != nullptr;
            const bool aux = m == 1 && !one_stream && !h->prof && h->imi_screen[0].ok && dc >= 16;
            TRY(screen_counters(h));
            DevBuf& b_xh = aux ? h->imi_ws2.xh : h->ws_xh;
            DevBuf& b_xflags = aux ? h->imi_ws2.xflags : h->ws_xflags;
            DevBuf& b_qn = aux ? h->imi_ws2.qn : h->ws_qn;
            DevBuf& b_qn_c = aux ? h->imi_ws2.qn_c : h->ws_qn_c;
            DevBuf& b_cand = aux ? h->imi_ws2.cand : h->ws_cand;
            DevBuf& b_tmin = aux ? h->imi_ws2.tmin : h->ws_tmin;
            float* subm = aux ? sub2 : sub;
            TRY(b_xh.reserve((size_t)n_pad * dp * 2));
            TRY(b_xflags.reserve((size_t)n));
            TRY(b_qn.reserve((size_t)n * 4));
            TRY(b_qn_c.reserve((size_t)n * 4));
            TRY(b_cand.reserve(vlq::coarse_screen_keep_bytes(n, kc)));
            if (kc > 8192 || vlq::coarse_screen_matrix_free_ok(kc, T)) TRY(b_tmin.reserve((size_t)n * (kc / 16 + 32) * sizeof(float)));
            hipStream_t st = h->stream;
            if (aux) {
                if (!h->imi_stream) {
                    HIP_TRY(hipStreamCreateWithFlags(&h->imi_stream, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&h->imi_fork, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&h->imi_join, hipEventDisableTiming));
                }
                st = h->imi_stream;
                HIP_TRY(hipStreamWaitEvent(st, h->imi_fork, 0));        // (recorded before half 0 was issued: inputs and workspace ready)
            } else if (m == 0 && !one_stream && !h->prof && h->imi_screen[1].ok) {
                if (!h->imi_stream) {
                    HIP_TRY(hipStreamCreateWithFlags(&h->imi_stream, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&h->imi_fork, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&h->imi_join, hipEventDisableTiming));
                }
                HIP_TRY(hipEventRecord(h->imi_fork, h->stream));
            }
            vlq::launch_gather_cols(x_dev, n, h->d, m * dc, dc, subm, st);
            vlq::launch_screen_prep(subm, sc.mu.as<float>(), n, dc, sc.scale, b_xh.p, b_qn.as<float>(), b_qn_c.as<float>(),
                                    b_xflags.as<unsigned char>(), st);
            vlq::launch_coarse_screened(subm, b_xh.p, b_xflags.as<unsigned char>(), cent, sc.half.p, b_qn.as<float>(),
                                        h->imi_norm.as<float>() + (size_t)m * kc, b_qn_c.as<float>(), sc.norm_c.as<float>(), tab[m],
                                        b_tmin.p ? b_tmin.as<float>() : nullptr, b_cand.p, n, kc, dc, T, sc.scale, sc.cmax, sc.cmax0, sv[m], si[m], nullptr,
                                        h->ws_screen_cnt.as<unsigned int>(), st);
            if (aux) {
                HIP_TRY(hipEventRecord(h->imi_join, st));
                HIP_TRY(hipStreamWaitEvent(h->stream, h->imi_join, 0));
            }
            screened_rows += n;
            continue;
        }
        if (dc < 16) {
            // compute_distance_table (ProductQuantizer.cpp:410-422): fvec_L2sqr per entry
            vlq::launch_gather_cols(x_dev, n, h->d, m * dc, dc, sub, h->stream);
            vlq::launch_pq_tables(sub, n, dc, cent, 1, kc, dc, nullptr, 1, tab[m], h->stream);
        } else {
            // pairwise_L2sqr (utils.cpp:1311-1355): (|x|^2 + |y|^2) - 2 <x,y>
            vlq::launch_gather_cols(x_dev, n, h->d, m * dc, dc, sub, h->stream);
            TRY(h->ws_qn.reserve((size_t)n * 4));
            vlq::launch_row_norms(sub, n, dc, h->ws_qn.as<float>(), h->stream);
            argmin = T == 1 && vlq::coarse_argmin_ok(kc, dc);
            if (argmin) {
                TRY(h->ws_tmin.reserve((size_t)n * (kc / 64) * 8));
                tmin = h->ws_tmin.as<float>();
            } else if (vlq::coarse_tile_minima_ok(kc, dc, T)) {
                TRY(h->ws_tmin.reserve((size_t)n * (kc / 64) * sizeof(float)));
                tmin = h->ws_tmin.as<float>();
            }
            vlq::launch_coarse_distances(sub, cent, h->ws_qn.as<float>(), h->imi_norm.as<float>() + (size_t)m * kc,
                                         argmin ? nullptr : tab[m], n, kc, dc, h->stream, tmin, argmin ? 0 : n_pad);
        }
        if (argmin) vlq::launch_coarse_argmin(tmin, n, kc, sv[m], si[m], h->stream);
        else if (T >= radix_from && vlq::row_select_sorted_ok(kc, T)) vlq::launch_row_select_sorted(tab[m], n, kc, kc, T, sv[m], si[m], h->stream);   // (imi_wide.hip)
        else vlq::launch_coarse_select(tab[m], n, kc, T, sv[m], si[m], h->stream, tmin);
    }
    if (screened_rows > 0) TRY(screen_counters_copy(h, screened_rows));      // (both chains have joined the index's stream)
    vlq::launch_imi_minsum(sv[0], si[0], sv[1], si[1], T, n, k, kc, h->imi_nbits, hv, hi, cdis_dev, keys_dev,
                           h->stream);
    HIP_TRY(hipGetLastError());
    return VLQ_OK;
}

int coarse_dev(vlq_ivfpq_t h, int64_t n, const float* x_dev, int nprobe, float* cdis_dev,
               int64_t* keys_dev) {
    StageTimer tm(h, 0);
    if (h->imi_nbits > 0) {
        const int64_t kc = int64_t(1) << h->imi_nbits;
        const int64_t page = std::max<int64_t>(1, std::min<int64_t>(32768, (int64_t)((size_t(1) << 29) / (size_t)kc)));
        for (int64_t i0 = 0; i0 < n; i0 += page) {
            const int64_t ni = std::min(page, n - i0);
            TRY(imi_page(h, ni, x_dev + i0 * h->d, nprobe, cdis_dev + i0 * nprobe, keys_dev + i0 * nprobe));
        }
        tm.stop();
        return VLQ_OK;
    }
    // knn_L2sqr dispatch (utils.cpp:935-946): small batches bypass the GEMM formulation
    const bool direct = (h->d % 4 == 0) && n < 20;
    // a 1-NN assignment writes no distance matrix: full pages whatever nlist is
    const int64_t page = (nprobe == 1 && !direct && vlq::coarse_argmin_ok(h->nlist, h->d)) ? 32768 : query_page(h);
    for (int64_t i0 = 0; i0 < n; i0 += page) {
        const int64_t ni = std::min(page, n - i0);
        TRY(coarse_page(h, ni, x_dev + i0 * h->d, nprobe, cdis_dev + i0 * nprobe,
                        keys_dev + i0 * nprobe, false, direct));
    }
    tm.stop();
    return VLQ_OK;
}

int scan_dev(vlq_ivfpq_t h, int64_t n, const float* x_dev, const int64_t* keys_dev,
             const float* cdis_dev, int nprobe, int k, float* D_dev, int64_t* I_dev,
             int store_pairs) {
    TRY(ensure_term2(h));
    const size_t E = (size_t)h->M * h->ksub;
    const int table_mode = !h->by_residual ? 2 : (h->use_precomputed_table == 1 ? 1 : 0);
    if (h->imi_nbits > 0 && table_mode == 0)
        return fail(VLQ_ERR_UNSUPPORTED, "multi-index coarse quantizer without the precomputed table (type 2) is not built");
    const int64_t page = 32768;
    // M=16 x 8 bit x d=128 in table mode 1: the scan kernel builds the per-query table itself
    // (and the 8 / 32 / 64-byte kernels of scanm.hip, any dsub, when they will serve the batch)
    const bool scanm_shape = table_mode == 1 && (h->M == 4 || h->M == 8 || h->M == 12 || (h->M >= 20 && h->M <= 32 && h->M % 4 == 0 && h->M != 16) ||
                                                 (h->M >= 40 && h->M <= 64 && h->M % 8 == 0)) && h->ksub == 256 &&
                             h->ntotal >= (int64_t)h->nlist * 24 && !getenv("VLQ_GENERIC_SCAN") && !getenv("VLQ_SCANM_QTAB");
    const bool fused_tables = (table_mode == 1 && h->M == 16 && h->ksub == 256 && h->dsub == 8) || scanm_shape;
    if (table_mode != 0 && !fused_tables)
        TRY(h->ws_qtab.reserve((size_t)std::min(n, page) * E * sizeof(float)));
    for (int64_t i0 = 0; i0 < n; i0 += page) {
        const int64_t ni = std::min(page, n - i0);
        const float* xi = x_dev + i0 * h->d;
        if (table_mode != 0 && !fused_tables) {
            StageTimer tm(h, 1);
            // init_query_L2 (IndexIVFPQ.cpp:557-563): ip table (mode 1) or distance table
            vlq::launch_pq_tables(xi, ni, h->d, h->pq.as<float>(), h->M, h->ksub, h->dsub, nullptr,
                                  table_mode == 1 ? 0 : 1, h->ws_qtab.as<float>(), h->stream);
            tm.stop();
        }
        vlq::ScanArgs a;
        a.codes = h->codes.as<uint8_t>();
        a.ids = h->ids.as<int64_t>();
        a.list_off = h->list_off.as<int64_t>();
        a.list_len = h->list_len.as<int64_t>();
        a.term2 = table_mode == 1 ? h->term2.as<float>() : nullptr;
        a.qtab = (table_mode != 0 && !fused_tables) ? h->ws_qtab.as<float>() : nullptr;
        a.queries = xi;
        a.coarse = h->coarse.as<float>();
        a.pq_cent = h->pq.as<float>();
        a.pq_cent_t = h->pq_t.as<float>();
        a.keys = keys_dev + i0 * nprobe;
        a.coarse_dis = cdis_dev + i0 * nprobe;
        a.D = D_dev + i0 * k;
        a.I = I_dev + i0 * k;
        a.ncode = h->stats.as<unsigned long long>();
        a.bad_key = reinterpret_cast<int*>(h->stats.as<unsigned long long>() + 1);
        a.nq = ni;
        a.nprobe = nprobe; a.k = k; a.M = h->M; a.ksub = h->ksub; a.dsub = h->dsub; a.d = h->d;
        a.nlist = h->nlist;
        a.table_mode = table_mode;
        a.imi_nbits = h->imi_nbits;
        a.max_codes = h->max_codes;
        a.store_pairs = store_pairs;
        // walking order of a query's probes (walk_order.cuh; speed only): the nearest probe first, the rest in list-id order
        // for the batches whose workgroups compete for the fabric -- k <= 64 (longer selections pay more for the late
        // admission bound than the rows save: k = 100 0.81 -> 0.84 ms), nprobe >= 16, 8- (two-wave shape, from 3000 queries on:
        // 0.366 -> 0.343 ms on the headline data, 2.15 -> 1.70 GB fetched; four waves 0.356 -> 0.359), every engineered size from
        // 12 to 56 bytes (12 / 20 / 24 / 28 / 40 / 48 / 56 bytes: 0.53 / 0.92 / 1.14 / 1.37 / 1.96 / 2.37 / 2.80 -> 0.47 / 0.82 /
        // 1.00 / 1.23 / 1.78 / 2.13 / 2.55 ms forced, more with the measured clock period; 64-byte codes 3.61 -> 3.47), and only when the batch's
        // neighbours share few lists
        // (walk_stat_kernel below).  VLQ_WALK_FIRST = n forces n probes in front for every batch, -1 the reference's order.
        static const int wf_env = [] { const char* e = getenv("VLQ_WALK_FIRST"); return e ? atoi(e) : -2; }();
        const bool walk_base = table_mode == 1 && h->imi_nbits == 0 &&
                               ((h->M >= 12 && h->M <= 64 && h->M % 4 == 0) || (h->M == 8 && ni >= 3000)) &&
                               h->ksub == 256 && ni >= 1024 && !h->fp16_tables;
        // k <= 64 from 16 probes on; 64 < k <= 128 from 64 probes on with the 4 nearest in front (headline data, nprobe 64,
        // k 100: 1.51 -> 1.38 ms; at nprobe 32 nothing to gain: 0.81 = 0.81) on indexes of short lists
        const int walk_rule = !walk_base ? -1 : (k <= 64 && nprobe >= 16) ? 1
                            : (k <= 128 && nprobe >= 64 && h->ntotal < (int64_t)h->nlist * 1024) ? 4 : -1;
        a.walk_first = wf_env >= -1 ? wf_env : walk_rule;
        {
            static const int wc = [] { const char* e = getenv("VLQ_WALK_CLOCK"); return e ? atoi(e) : 0; }();       // > 0 fixed period, < 0 no clock
            static const int wscale = [] { const char* e = getenv("VLQ_WALK_SCALE"); return e ? atoi(e) : 1000; }();
            a.walk_clock = wc > 0 ? wc : 0;
            a.walk_scale = wscale;
            if (wc == 0 && a.walk_first >= 0) {
                // the workgroups' own walk times, per XCD; a new (nprobe, k, batch class) starts measuring afresh
                if (!h->walk_state.p) { TRY(h->walk_state.reserve(8 * 16 * sizeof(int))); h->walk_key = -1; }
                const int64_t wkey = ((int64_t)nprobe << 32) ^ ((int64_t)k << 16) ^ (int64_t)(ni >= 4096 ? 2 : 1);
                if (wkey != h->walk_key) { (void)hipMemsetAsync(h->walk_state.p, 0, 8 * 16 * sizeof(int), h->stream); h->walk_key = wkey; h->walk_stat_calls = 0; }
                a.walk_state = h->walk_state.as<int>();
            }
        }
        // the statistic is computed with the scan order (launch_query_order); behind the order's ni entries: its 32 counts
        const bool walk_auto = wf_env < -1 && a.walk_first >= 0;
        // (the counts live in the handle: the statistic describes the workload, not one batch -- it is sampled on the first
        // four searches of a (nprobe, k, batch class) and on every 16th after that, 6.4 us + a launch gap otherwise saved per
        // search; VLQ_WALK_STAT_EVERY=1: every search.  Speed only: the results do not depend on the walking order)
        if (walk_auto) TRY(h->walk_counts.reserve(32 * sizeof(int)));
        static const int stat_every = [] { const char* e = getenv("VLQ_WALK_STAT_EVERY"); return e ? std::max(1, atoi(e)) : 16; }();
        const bool walk_stat_now = walk_auto && (h->walk_stat_calls < 4 || h->walk_stat_calls % stat_every == 0);
        if (walk_auto) h->walk_stat_calls++;
        auto walk_part = [&]() -> int* { return walk_auto ? h->walk_counts.as<int>() : nullptr; };
        // a launch with no measured walk time seeds its clock period from a model (walk_stat_kernel): the workgroups that will
        // share the chip = the scan kernels' slots (scan16.hip: 2048 two-wave / 1280 four-wave workgroups), at most the batch
        vlq::WalkSeed wseed;
        wseed.list_off = h->list_off.as<int64_t>(); wseed.list_len = h->list_len.as<int64_t>(); wseed.nlist = h->nlist;
        wseed.slots = (int)std::min<int64_t>(ni, (k <= 128 && ni >= 3000 && h->ntotal < (int64_t)h->nlist * 1024) ? 2048 : (k <= 64 ? 1280 : 1024));
        auto walk_decide = [&]() {        // after launch_query_order
            if (!walk_auto || !a.qorder) return;
            static const int share_max = [] { const char* e = getenv("VLQ_WALK_SHARE"); return e ? atoi(e) : 300; }();
            const int samples = vlq::walk_stat_samples(ni, nprobe);
            // from 128 probes on the list-id order won on both data sets (G1 2.26 -> 1.97 ms, headline 3.02 -> 2.48)
            a.walk_limit = (int)((int64_t)samples * ((nprobe >= 128 && k <= 64) ? 1000 : share_max) / 1000);
            a.walk_flag = walk_part();
            if (getenv("VLQ_WALK_STAT_PRINT")) {
                int v[32], tot = 0;
                (void)hipStreamSynchronize(h->stream);
                (void)hipMemcpy(v, a.walk_flag, sizeof(v), hipMemcpyDeviceToHost);
                for (int x : v) tot += x;
                int ws[8 * 16] = {0};
                if (a.walk_state) (void)hipMemcpy(ws, a.walk_state, sizeof(ws), hipMemcpyDeviceToHost);
                fprintf(stderr, "[vlq] walk order: neighbours share %d of %d sampled probes -> %s; walk ticks per XCD %d %d %d %d %d %d %d %d\n", tot, samples,
                        tot <= a.walk_limit ? "list-id order" : "coarse-distance order", ws[0], ws[16], ws[32], ws[48], ws[64], ws[80], ws[96], ws[112]);
            }
        };
        const bool fast16 = table_mode == 1 && h->M == 16 && h->ksub == 256;
        if (h->fp16_tables && fast16 && h->imi_nbits == 0 && k <= 256) {
            // useFloat16LookupTables: half(term 2) once per trained state, half(term 3) per page, half table sums
            TRY(ensure_term2h(h));
            TRY(h->ws_qtab.reserve((size_t)ni * E * sizeof(float)));
            TRY(h->ws_qtabh.reserve((size_t)ni * E * 2));
            {
                StageTimer tq(h, 1);
                vlq::launch_pq_tables(xi, ni, h->d, h->pq.as<float>(), h->M, h->ksub, h->dsub, nullptr, 0,
                                      h->ws_qtab.as<float>(), h->stream);
                vlq::launch_to_half(h->ws_qtab.as<float>(), ni * (int64_t)E, -2.f, h->ws_qtabh.as<uint16_t>(), h->stream);
                if (ni >= 1024) {
                    TRY(h->ws_hist.reserve(2 * vlq::query_order_bins_padded(h->nlist) * sizeof(int)));
                    TRY(h->ws_qorder.reserve(((size_t)ni + 40) * sizeof(int)));
                    vlq::launch_query_order(a.keys, ni, nprobe, h->nlist, h->ws_hist.as<int>(), h->ws_qorder.as<int>(), h->stream,
                                            h->have_rank ? h->list_rank.as<int>() : nullptr);
                    a.qorder = h->ws_qorder.as<int>();
                }
                tq.stop();
            }
            a.term2h = h->term2h.as<uint16_t>();
            a.qtabh = h->ws_qtabh.as<uint16_t>();
            StageTimer tm(h, 2);
            vlq::launch_scan16h(a, h->stream);
            tm.stop();
            continue;
        }
        a.long_lists = h->ntotal >= (int64_t)h->nlist * 1024;   // mean list >= 4 chunks of 256 codes
        if (fast16) {
            // scan schedule (speed only): list-owned = one workgroup per (query, list partition), XCD x
            // serves the lists of partition x, so their term2 rows and codes stay in that XCD's L2
            const int sched = h->scan_schedule ? h->scan_schedule : 1;     // 0 = automatic = query-major (the faster one on every data set measured)
            const bool owned = sched >= 2 && h->imi_nbits == 0 && h->have_rank && h->nlist >= 64 && h->nlist <= 16384 &&
                               ni >= 1024 && nprobe >= 8 && h->dsub == 8 && h->ntotal >= (int64_t)h->nlist * 24;
            if (owned) {
                TRY(h->ws_own_hist.reserve(vlq::owned_hist_ints(h->nlist) * sizeof(int)));
                TRY(h->ws_own_minr.reserve((size_t)ni * 8 * sizeof(int)));
                TRY(h->ws_own_order.reserve((size_t)ni * 8 * sizeof(int)));
                TRY(h->ws_own_count.reserve(64));
                TRY(h->ws_part_mask.reserve((size_t)ni + 16));
                TRY(h->ws_part_keys.reserve((size_t)ni * 8 * k * 8));
                TRY(h->ws_qtab.reserve((size_t)ni * E * sizeof(float)));
                if (sched >= 3 && vlq::scan16o_supports(a)) {         // second build: per-probe records, 8-byte item entries
                    TRY(h->ws_own_recs.reserve((size_t)ni * nprobe * sizeof(vlq::OwnRec)));
                    TRY(h->ws_own_seg.reserve((size_t)ni * 8 * 4));
                    TRY(h->ws_own_items.reserve((size_t)ni * 8 * 8));
                    vlq::ScanArgs ao = a;
                    ao.qorder = nullptr;
                    ao.qtab = h->ws_qtab.as<float>();
                    ao.qtab_scaled = 1;
                    ao.list_part = h->list_part.as<uint8_t>();
                    ao.own_count = h->ws_own_count.as<int>();
                    ao.part_mask = h->ws_part_mask.as<uint8_t>();
                    ao.part_keys = h->ws_part_keys.as<unsigned long long>();
                    ao.own_recs = h->ws_own_recs.as<vlq::OwnRec>();
                    ao.own_items = h->ws_own_items.as<uint2>();
                    {
                        StageTimer tq(h, 1);
                        vlq::launch_owned2_prepare(ao, h->list_rank.as<int>(), h->ws_own_hist.as<int>(), h->ws_own_minr.as<int>(),
                                                   h->ws_own_seg.as<uint32_t>(), h->ws_own_items.as<uint2>(), h->ws_own_count.as<int>(),
                                                   h->ws_part_mask.as<uint8_t>(), h->ws_own_recs.as<vlq::OwnRec>(), h->stream);
                        vlq::launch_qtab16(xi, ni, h->pq_t.as<float>(), h->ws_qtab.as<float>(), h->stream);
                        tq.stop();
                    }
                    if (getenv("VLQ_PHASE_TIMING")) {      // diagnostic: items per partition
                        static int once = 0;
                        if (!once++) {
                            int cnt[8];
                            (void)hipStreamSynchronize(h->stream);
                            (void)hipMemcpy(cnt, h->ws_own_count.p, 32, hipMemcpyDeviceToHost);
                            fprintf(stderr, "[owned] items per partition: %d %d %d %d %d %d %d %d\n", cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7]);
                        }
                    }
                    StageTimer tm(h, 2);
                    vlq::launch_scan16_owned2(ao, sched == 4 ? 2 : 1, h->stream);
                    vlq::launch_owned_merge(ao, h->stream);
                    tm.stop();
                    continue;
                }
                {
                    StageTimer tq(h, 1);   // item ordering + per-query tables are booked with the table stage
                    vlq::launch_owned_order(a.keys, ni, nprobe, h->nlist, h->list_rank.as<int>(), h->list_part.as<uint8_t>(),
                                            h->ws_own_hist.as<int>(), h->ws_own_minr.as<int>(), h->ws_own_order.as<int>(),
                                            h->ws_own_count.as<int>(), h->ws_part_mask.as<uint8_t>(), h->stream);
                    vlq::launch_qtab16(xi, ni, h->pq_t.as<float>(), h->ws_qtab.as<float>(), h->stream);
                    tq.stop();
                }
                vlq::ScanArgs ao = a;
                ao.qorder = nullptr;
                ao.qtab = h->ws_qtab.as<float>();
                ao.qtab_scaled = 1;
                ao.list_part = h->list_part.as<uint8_t>();
                ao.own_order = h->ws_own_order.as<int>();
                ao.own_count = h->ws_own_count.as<int>();
                ao.part_mask = h->ws_part_mask.as<uint8_t>();
                ao.part_keys = h->ws_part_keys.as<unsigned long long>();
                StageTimer tm(h, 2);   // the scan of the items + the join of a query's parts
                vlq::launch_scan16_owned(ao, h->stream);
                vlq::launch_owned_merge(ao, h->stream);
                tm.stop();
                continue;
            }
            if (ni >= 1024 && h->nlist <= (1 << 22)) {
                StageTimer tq(h, 1);   // query ordering is booked with the table stage
                // run queries that share their nearest centroid next to each other (L2 reuse)
                TRY(h->ws_hist.reserve(2 * vlq::query_order_bins_padded(h->nlist) * sizeof(int)));
                TRY(h->ws_qorder.reserve(((size_t)ni + 40) * sizeof(int)));
                vlq::launch_query_order(a.keys, ni, nprobe, h->nlist, h->ws_hist.as<int>(),
                                        h->ws_qorder.as<int>(), h->stream,
                                        (h->have_rank && h->imi_nbits == 0) ? h->list_rank.as<int>() : nullptr, walk_part(), a.walk_state, wseed, walk_stat_now, h->order_hist_ready && ni == n);
                a.qorder = h->ws_qorder.as<int>();
                walk_decide();
                tq.stop();
            }
            StageTimer tm(h, 2);       // exactly the scan kernel
            if (h->ntotal < (int64_t)h->nlist * 24) {       // a few codes per list
                vlq::launch_scan16_short(a, h->stream);
                snprintf(h->last_scan, sizeof(h->last_scan), "scan16_short_kernel");
            }
            else {
                // fewer workgroups than the chip holds (256 CUs x 4): split every query's probes over
                // several workgroups and join the partial rows -- serving-size batches
                // (mid-size batches -- 1250 / 2500 queries, the slices of a batch sharded over 8 / 4 GPUs, which
                // fill the 1024 slots a fractional number of times -- were tried with 2-8 parts too: a workgroup
                // costs about 19 us of slot time before and after its probes against 1.5 us per probe, so the
                // finer granularity buys nothing: 625 queries 0.090 -> 0.123 ms split in 8, 1250 queries 0.16 ms
                // either way; tools/slice_stages.py)
                int nsplit = 1;
                while (k <= 256 && nsplit < 8 && ni * nsplit * 2 <= 1024 && nprobe / (nsplit * 2) >= 4) nsplit *= 2;
                if (nsplit > 1) {
                    TRY(h->ws_Dp.reserve((size_t)nsplit * ni * k * sizeof(float)));
                    TRY(h->ws_Ip.reserve((size_t)nsplit * ni * k * sizeof(int64_t)));
                    vlq::ScanArgs ap = a;
                    ap.nsplit = nsplit;
                    ap.D = h->ws_Dp.as<float>();
                    ap.I = h->ws_Ip.as<int64_t>();
                    vlq::launch_scan16(ap, h->stream);
                    vlq::launch_merge_topk(ap.D, ap.I, ni, k, nsplit, a.D, a.I, h->stream);
                } else if (k > 256 || (k > 128 && !a.long_lists)) {
                    // one selection per workgroup.  128 < k <= 256 (round 3, 10 000 queries): bench index (lists of ~700
                    // codes where probed) k = 200 1.06 -> 0.89 ms, k = 256 1.14 -> 0.90 ms against the per-wave lists of
                    // scan16_kernel<4>; lists of 3 906 codes 4.01 / 4.14 ms for the pipelined scan16 against 4.56 / 4.59:
                    // the trip barriers of the shared queue cost more than four private merge networks there
                    vlq::launch_scan16_bigk(a, h->stream);
                } else {
                    // A batch that fills the chip's 4 x #CU workgroup slots a fractional number of times leaves most of the
                    // chip idle in its last round (1250 queries, the slice of a 10 000-query batch on one of 8 GPUs: 1024 +
                    // 226): the queries of that last round are split into parts (kernels.h: tail_r / tail_p), so that it is a
                    // round of SHORT workgroups.  Measured (scan stage, G1 / headline data): 1100 queries 0.118 -> 0.103 / 0.139 ->
                    // 0.125 ms, 1250 queries 0.122 -> 0.118 / 0.146 -> 0.133; from the third round on (2500 queries) it no longer
                    // pays -- workgroups of an under-filled chip run faster as it is -- so only the second round is split
                    const int64_t slots = (k <= 64 && !a.long_lists && h->imi_nbits == 0) ? 1280 : 1024;   // workgroups the chip holds (scan16.hip)
                    const int64_t rem = ni % slots;
                    int tp = rem > 0 ? (int)std::min<int64_t>(8, slots / rem) : 1;
                    tp = std::min(tp, nprobe / 4);
                    static const bool tail_off = getenv("VLQ_NO_TAIL_SPLIT") != nullptr;
                    if (ni > slots && ni < 2 * slots && tp >= 2 && k <= 128 && !tail_off) {
                        vlq::ScanArgs at = a;
                        at.tail_r = (int)((rem + 7) / 8);
                        at.tail_p = tp;
                        const size_t rows = (size_t)8 * at.tail_r;
                        TRY(h->ws_Dp.reserve((size_t)tp * rows * k * sizeof(float)));
                        TRY(h->ws_Ip.reserve((size_t)tp * rows * k * sizeof(int64_t)));
                        TRY(h->ws_misc.reserve(rows * sizeof(int)));
                        HIP_TRY(hipMemsetAsync(h->ws_misc.p, 0xFF, rows * sizeof(int), h->stream));
                        at.tail_D = h->ws_Dp.as<float>();
                        at.tail_I = h->ws_Ip.as<int64_t>();
                        at.tail_rows = h->ws_misc.as<int>();
                        vlq::launch_scan16(at, h->stream);
                        vlq::launch_merge_topk(at.tail_D, at.tail_I, (int64_t)rows, k, tp, a.D, a.I, h->stream, at.tail_rows);
                    } else {
                        vlq::launch_scan16(a, h->stream);
                    }
                }
            }
            tm.stop();
            if (h->ntotal >= (int64_t)h->nlist * 24) snprintf(h->last_scan, sizeof(h->last_scan), "%s", vlq::last_scan16_shape());
            h->last_walk_first = a.walk_first; h->last_walk_limit = a.walk_limit;
            h->last_walk_samples = a.walk_flag ? vlq::walk_stat_samples(ni, nprobe) : 0;
            h->last_walk_counts = a.walk_flag != nullptr;       // (the 32 counts the order was decided from live in the handle)
        } else if ((vlq::scanm_supports(a) || vlq::scanm0_supports(a)) && h->ntotal >= (int64_t)h->nlist * 24 && !getenv("VLQ_GENERIC_SCAN")) {
            // 8 / 32 / 64-byte codes: the engineered organisation (scanm.hip); queries ordered like the 16-byte path
            if (ni >= 1024 && h->nlist <= (1 << 22)) {
                StageTimer tq(h, 1);
                TRY(h->ws_hist.reserve(2 * vlq::query_order_bins_padded(h->nlist) * sizeof(int)));
                TRY(h->ws_qorder.reserve(((size_t)ni + 40) * sizeof(int)));
                vlq::launch_query_order(a.keys, ni, nprobe, h->nlist, h->ws_hist.as<int>(), h->ws_qorder.as<int>(), h->stream,
                                        (h->have_rank && h->imi_nbits == 0) ? h->list_rank.as<int>() : nullptr, walk_part(), a.walk_state, wseed, walk_stat_now, h->order_hist_ready && ni == n);
                a.qorder = h->ws_qorder.as<int>();
                walk_decide();
                tq.stop();
            }
            StageTimer tm(h, 2);
            vlq::launch_scanm(a, h->stream);
            tm.stop();
            snprintf(h->last_scan, sizeof(h->last_scan), "scanm_kernel<%d>", h->M);
            h->last_walk_first = a.walk_first; h->last_walk_limit = a.walk_limit;
            h->last_walk_samples = a.walk_flag ? vlq::walk_stat_samples(ni, nprobe) : 0;
            h->last_walk_counts = a.walk_flag != nullptr;
        } else if (h->M != 16 && h->ntotal < (int64_t)h->nlist * 24 && vlq::scanm_short_supports(a) && !getenv("VLQ_GENERIC_SCAN")) {
            // a few codes per list, any engineered code size but 16 bytes (the multi-index drivers ship 8): no table per probe,
            // each lane fetches the entries its code addresses (scanm_short.hip)
            StageTimer tm(h, 2);
            vlq::launch_scanm_short(a, h->stream);
            tm.stop();
            snprintf(h->last_scan, sizeof(h->last_scan), "scanm_short_kernel<%d>", h->M);
            h->last_walk_first = -1; h->last_walk_limit = 0; h->last_walk_samples = 0; h->last_walk_counts = false;
        } else {
            StageTimer tm(h, 2);
            vlq::launch_scan(a, h->stream);
            tm.stop();
            snprintf(h->last_scan, sizeof(h->last_scan), "scan_kernel");
            h->last_walk_first = -1; h->last_walk_limit = 0; h->last_walk_samples = 0; h->last_walk_counts = false;
        }
    }
    HIP_TRY(hipGetLastError());
    h->stat_nq += (uint64_t)n;
    h->order_hist_ready = false;
    return VLQ_OK;
}

int check_search_args(vlq_ivfpq_t h, int64_t n, const void* x, int nprobe, int k, const void* D,
                      const void* I) {
    if (n < 0) return fail(VLQ_ERR_INVALID, "n < 0");
    if (n > 0 && (!x || !D || !I)) return fail(VLQ_ERR_INVALID, "null buffer");
    // (a multi-index quantizer's coarse stage goes on to VLQ_MAX_IMI_NPROBE cells: imi_wide.hip)
    const int max_probe = (h && h->imi_nbits > 0) ? VLQ_MAX_IMI_NPROBE : VLQ_MAX_NPROBE;
    if (nprobe < 1 || nprobe > max_probe)
        return fail(VLQ_ERR_INVALID, "nprobe=%d outside 1..%d", nprobe, max_probe);
    if (k < 1 || k > VLQ_MAX_K) return fail(VLQ_ERR_INVALID, "k=%d outside 1..%d", k, VLQ_MAX_K);
    return VLQ_OK;
}

int finish_outputs(vlq_ivfpq_t h, bool copyD, void* D, const void* Dd, size_t bytesD, bool copyI,
                   void* I, const void* Id, size_t bytesI) {
    if (copyD) HIP_TRY(hipMemcpyAsync(D, Dd, bytesD, hipMemcpyDeviceToHost, h->stream));
    if (copyI) HIP_TRY(hipMemcpyAsync(I, Id, bytesI, hipMemcpyDeviceToHost, h->stream));
    if (copyD || copyI) HIP_TRY(hipStreamSynchronize(h->stream));
    return VLQ_OK;
}

// device flag word of h->stats: 1 = a probe key >= nlist (the caller's error), 2 = a scan kernel found static LDS
// in front of its look-up tables (a build fault of this library: the gathers use absolute LDS offsets)
int bad_flag_error(int bad) {
    if (bad == 2) return fail(VLQ_ERR_HIP, "internal: a 16-byte scan kernel was built with static LDS (its table offsets are absolute)");
    return fail(VLQ_ERR_INVALID, "a probe key >= nlist was passed to search_preassigned (IndexIVFPQ.cpp:1008-1011)");
}

// An out-of-range probe key aborts the reference's search (IndexIVFPQ.cpp:1008-1011).  The scan
// kernels raise a device flag; call this after the stream has been synchronised (host outputs).
int read_bad_key(vlq_ivfpq_t h) {
    int bad = 0;
    HIP_TRY(hipMemcpy(&bad, reinterpret_cast<const char*>(h->stats.p) + 8, sizeof(int), hipMemcpyDeviceToHost));
    if (!bad) return VLQ_OK;
    HIP_TRY(hipMemset(reinterpret_cast<char*>(h->stats.p) + 8, 0, 8));
    return bad_flag_error(bad);
}

}  // namespace

static vlq::ListStore list_store(vlq_ivfpq_t h) {
    vlq::ListStore ls;
    ls.nlist = h->nlist; ls.code_size = h->M;
    ls.codes = &h->codes; ls.ids = &h->ids; ls.off = &h->list_off; ls.len = &h->list_len;
    ls.h_off = &h->h_list_off; ls.h_len = &h->h_list_len; ls.h_stale = &h->h_lists_stale;
    return ls;
}

extern "C" {

int vlq_version(void) { return 100; }

const char* vlq_last_error(void) { return err_slot().c_str(); }

int vlq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int vlq_ivfpq_create(vlq_ivfpq_t* out, int device, int d, int nlist, int M, int nbits) {
    if (!out) return fail(VLQ_ERR_INVALID, "null out");
    *out = nullptr;
    if (d <= 0 || nlist <= 0 || M <= 0) return fail(VLQ_ERR_INVALID, "d, nlist, M must be positive");
    if (d % M != 0) return fail(VLQ_ERR_INVALID, "d=%d not a multiple of M=%d", d, M);   // ProductQuantizer.cpp:165
    if (nbits < 1 || nbits > 8) return fail(VLQ_ERR_INVALID, "nbits=%d outside 1..8", nbits);  // IndexIVFPQ.cpp:51
    if ((size_t)M * (size_t)(1 << nbits) * 4 > 144 * 1024)
        return fail(VLQ_ERR_UNSUPPORTED, "M * 2^nbits lookup table exceeds the LDS budget");
    int ndev = vlq_device_count();
    if (ndev <= 0) return fail(VLQ_ERR_HIP, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(VLQ_ERR_INVALID, "device %d out of range", device);
    vlq_ivfpq_s* h = new (std::nothrow) vlq_ivfpq_s();
    if (!h) return fail(VLQ_ERR_INVALID, "out of memory");
    h->device = device; h->d = d; h->nlist = nlist; h->M = M; h->nbits = nbits;
    h->ksub = 1 << nbits; h->dsub = d / M;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(VLQ_ERR_HIP, "device init failed: %s", hipGetErrorString(e)); }
    h->stream = h->own_stream;
    if (const char* e = getenv("VLQ_SCAN_SCHEDULE")) {    // tests / A-B runs: 1 query-major, 2 list-owned; anything else is ignored
        const int m = atoi(e);
        if (m >= 0 && m <= 4) h->scan_schedule = m;
    }
    if (const char* e = getenv("VLQ_COARSE_SCREEN")) h->coarse_screen = atoi(e);   // 0: f32 MFMA matrix path everywhere (A/B)
    if (const char* e = getenv("VLQ_COARSE_FILTER")) h->coarse_filter = atoi(e);   // 1: filtered coarse stage (A/B; slower)
    h->h_lists_stale = true;    // host copies of the list starts / lengths are filled on first use
    int rc = h->stats.reserve(512);    // [0] ncode, [1] flag word; [2..7] phase clocks of instrumented builds (-DVLQ_PHASE_TIMING), [8..47] (-DVLQ_SCAN16_PHASES)
    if (rc == VLQ_OK) rc = h->list_off.reserve(((size_t)nlist + 1) * 8);
    if (rc == VLQ_OK) rc = h->list_len.reserve((size_t)nlist * 8);
    if (rc == VLQ_OK) rc = h->codes.reserve(16);
    if (rc == VLQ_OK) rc = h->ids.reserve(16);
    if (rc != VLQ_OK) { vlq_ivfpq_destroy(h); return rc; }
    (void)hipMemsetAsync(h->stats.p, 0, 512, h->stream);
    (void)hipMemsetAsync(h->list_off.p, 0, ((size_t)nlist + 1) * 8, h->stream);
    (void)hipMemsetAsync(h->list_len.p, 0, (size_t)nlist * 8, h->stream);
    (void)hipStreamSynchronize(h->stream);
    h->have_lists = true;   // an empty index is searchable (all lists empty)
    *out = h;
    return VLQ_OK;
}

void vlq_ivfpq_destroy(vlq_ivfpq_t h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    drain_profile(h);
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    DevBuf* bufs[] = {&h->term2h, &h->ws_qtabh, &h->coarse, &h->cnorm, &h->pq, &h->pq_t, &h->rnorm, &h->term2, &h->codes, &h->ids,
                      &h->list_off, &h->list_len, &h->list_rank, &h->list_part, &h->ws_own_hist, &h->ws_own_minr, &h->ws_own_order,
                      &h->ws_own_count, &h->ws_part_mask, &h->ws_part_keys, &h->ws_own_recs, &h->ws_own_seg, &h->ws_own_items, &h->coarse_s, &h->cnorm_s, &h->ws_cand, &h->ws_cnt, &h->ws_Dp, &h->ws_Ip, &h->ws_append.cnt, &h->ws_append.cstart, &h->ws_append.keys_in,
                      &h->ws_append.keys_out, &h->ws_append.sort_tmp, &h->ws_x, &h->ws_qn, &h->ws_dist, &h->ws_keys, &h->ws_cdis,
                      &h->ws_qtab, &h->ws_D, &h->ws_I, &h->ws_misc, &h->ws_keys_in, &h->ws_cdis_in,
                      &h->ws_codes, &h->ws_assign, &h->ws_hist, &h->ws_qorder, &h->ws_tmin, &h->walk_state, &h->stats, &h->imi_cent, &h->ws_Dr, &h->ws_Ir, &h->ws_keys_run, &h->ws_cdis_run, &h->walk_counts,
                      &h->imi_norm, &h->imi_virtual, &h->ws_imi,
                      // the float16 screen of the coarse stage: built for every index at set_coarse_centroids
                      &h->screen.half, &h->screen.mu, &h->screen.norm_c, &h->imi_screen[0].half, &h->imi_screen[0].mu,
                      &h->imi_screen[0].norm_c, &h->imi_screen[1].half, &h->imi_screen[1].mu, &h->imi_screen[1].norm_c,
                      &h->ws_qn_c, &h->ws_xh, &h->ws_xflags, &h->ws_screen_cnt};
    for (auto b : bufs) b->release();
    if (h->screen_cnt_host) (void)hipHostFree(h->screen_cnt_host);
    if (h->ws_kept.p) {
        unsigned long long kept = 0;
        (void)hipMemcpy(&kept, h->ws_kept.p, 8, hipMemcpyDeviceToHost);
        fprintf(stderr, "[vlq] coarse screen: %llu columns kept in total\n", kept);
        h->ws_kept.release();
    }
    for (DevBuf* b : {&h->imi_ws2.xh, &h->imi_ws2.xflags, &h->imi_ws2.qn, &h->imi_ws2.qn_c, &h->imi_ws2.cand, &h->imi_ws2.tmin}) b->release();
    if (h->imi_fork) (void)hipEventDestroy(h->imi_fork);
    if (h->imi_join) (void)hipEventDestroy(h->imi_join);
    if (h->imi_stream) (void)hipStreamDestroy(h->imi_stream);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int vlq_ivfpq_set_stream(vlq_ivfpq_t h, void* hip_stream) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    TRY(set_dev(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);   // NULL = the HIP null stream
    return VLQ_OK;
}

// Spatial order of the lists (speed only): recursive two-means bisection of the centroids; lists
// that are close in space get close ranks.  The scan runs queries in the order of the rank of
// their nearest list, so workgroups that are resident together on an XCD probe neighbouring
// lists and find each other's term2 rows in L2.
static void spatial_list_rank(const float* cent, int nlist, int d, std::vector<int>& rank) {
    std::vector<int> order((size_t)nlist);
    for (int i = 0; i < nlist; i++) order[(size_t)i] = i;
    std::vector<float> ca((size_t)d), cb((size_t)d);
    std::vector<double> sa((size_t)d), sb((size_t)d);
    std::vector<char> side;
    struct Seg { int lo, hi; };
    std::vector<Seg> stack;
    stack.push_back({0, nlist});
    auto dist2 = [&](const float* x, const float* c) {
        float s = 0.f;
        for (int j = 0; j < d; j++) { const float t = x[j] - c[j]; s += t * t; }
        return s;
    };
    while (!stack.empty()) {
        const Seg sg = stack.back();
        stack.pop_back();
        const int n = sg.hi - sg.lo;
        if (n <= 2) continue;
        int* idx = order.data() + sg.lo;
        // two far-apart seeds: the point farthest from the first one, then the farthest from that
        const float* p0 = cent + (size_t)idx[0] * d;
        int fb = 0; float best = -1.f;
        for (int i = 0; i < n; i++) { const float v = dist2(cent + (size_t)idx[i] * d, p0); if (v > best) { best = v; fb = i; } }
        std::copy(cent + (size_t)idx[fb] * d, cent + (size_t)idx[fb] * d + d, cb.begin());
        int fa = 0; best = -1.f;
        for (int i = 0; i < n; i++) { const float v = dist2(cent + (size_t)idx[i] * d, cb.data()); if (v > best) { best = v; fa = i; } }
        std::copy(cent + (size_t)idx[fa] * d, cent + (size_t)idx[fa] * d + d, ca.begin());
        side.assign((size_t)n, 0);
        int na = 0;
        for (int it = 0; it < 4; it++) {
            std::fill(sa.begin(), sa.end(), 0.0);
            std::fill(sb.begin(), sb.end(), 0.0);
            na = 0;
            for (int i = 0; i < n; i++) {
                const float* x = cent + (size_t)idx[i] * d;
                const bool toa = dist2(x, ca.data()) < dist2(x, cb.data());
                side[(size_t)i] = toa;
                std::vector<double>& acc = toa ? sa : sb;
                for (int j = 0; j < d; j++) acc[(size_t)j] += x[j];
                na += toa;
            }
            if (na == 0 || na == n) break;
            for (int j = 0; j < d; j++) { ca[(size_t)j] = (float)(sa[(size_t)j] / na); cb[(size_t)j] = (float)(sb[(size_t)j] / (n - na)); }
        }
        if (na == 0 || na == n) continue;        // duplicates: leave the segment as it is
        // stable partition: side a first
        std::vector<int> tmp((size_t)n);
        int pa = 0, pb = na;
        for (int i = 0; i < n; i++) tmp[(size_t)(side[(size_t)i] ? pa++ : pb++)] = idx[i];
        std::copy(tmp.begin(), tmp.end(), idx);
        stack.push_back({sg.lo, sg.lo + na});
        stack.push_back({sg.lo + na, sg.hi});
    }
    rank.assign((size_t)nlist, 0);
    for (int i = 0; i < nlist; i++) rank[(size_t)order[(size_t)i]] = i;
}

// float16 screen of a coarse stage (coarse_screen.hip) for one centroid set: the centroids' mean, power-of-two scale from the
// largest centred |component|, largest centred / uncentred norm (rounded up), half copy and centred norms on the device.
// hc: host copy of the n x d centroids at cent_dev.
static int build_screen(vlq_ivfpq_t h, const float* hc, const float* cent_dev, int n, int d, vlq_ivfpq_s::ScreenSet& sc) {
    sc.ok = false;
    if (d > 128 || n < 1) return VLQ_OK;
    std::vector<double> mud((size_t)d, 0.0);
    bool finite = true;
    for (int i = 0; i < n; i++)
        for (int c = 0; c < d; c++) {
            const double v = hc[(size_t)i * d + c];
            finite = finite && std::isfinite(v);
            mud[(size_t)c] += v;
        }
    if (!finite) return VLQ_OK;
    std::vector<float> mu((size_t)d);
    for (int c = 0; c < d; c++) mu[(size_t)c] = (float)(mud[(size_t)c] / n);
    double amax = 0.0, nmax = 0.0, nmax0 = 0.0;
    for (int i = 0; i < n; i++) {
        double nn = 0.0, n0 = 0.0;
        for (int c = 0; c < d; c++) {
            const double v0 = hc[(size_t)i * d + c];
            const double v = (double)(float)(hc[(size_t)i * d + c] - mu[(size_t)c]);     // fl(c - mu), as the kernels form it
            amax = std::max(amax, std::fabs(v));
            nn += v * v;
            n0 += v0 * v0;
        }
        nmax = std::max(nmax, nn);
        nmax0 = std::max(nmax0, n0);
    }
    if (!(amax > 0.0 && amax < 1e30)) return VLQ_OK;
    int e = 0;
    (void)std::frexp(16384.0 / amax, &e);            // 16384 / amax = m * 2^e, m in [0.5, 1)
    sc.scale = std::ldexp(1.f, std::max(-100, std::min(100, e - 1)));     // s * amax <= 16384
    sc.cmax = (float)(std::sqrt(nmax) * 1.0001);
    sc.cmax0 = (float)(std::sqrt(nmax0) * 1.0001);
    const int dp = (d + 15) / 16 * 16;
    TRY(sc.mu.reserve((size_t)d * sizeof(float)));
    HIP_TRY(hipMemcpy(sc.mu.p, mu.data(), (size_t)d * sizeof(float), hipMemcpyHostToDevice));
    TRY(sc.half.reserve((size_t)((n + 127) / 128 * 128) * dp * 2));
    TRY(sc.norm_c.reserve((size_t)n * sizeof(float)));
    TRY(h->ws_misc.reserve((size_t)n * sizeof(float)));
    vlq::launch_screen_prep(cent_dev, sc.mu.as<float>(), n, d, sc.scale, sc.half.p, h->ws_misc.as<float>(), sc.norm_c.as<float>(), nullptr,
                            h->stream);
    HIP_TRY(hipStreamSynchronize(h->stream));
    sc.ok = true;
    return VLQ_OK;
}

int vlq_ivfpq_set_coarse_centroids(vlq_ivfpq_t h, const float* centroids) {
    if (!h || !centroids) return fail(VLQ_ERR_INVALID, "null argument");
    TRY(set_dev(h));
    const size_t bytes = (size_t)h->nlist * h->d * sizeof(float);
    TRY(h->coarse.reserve(bytes));
    TRY(h->cnorm.reserve((size_t)h->nlist * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(h->coarse.p, centroids, bytes, hipMemcpyDefault, h->stream));
    // y_norms of knn_L2sqr_blas (utils.cpp:857-858), computed once
    vlq::launch_row_norms(h->coarse.as<float>(), h->nlist, h->d, h->cnorm.as<float>(), h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->have_rank = false;
    h->screen.ok = false;
    if (h->nlist <= (1 << 17)) {      // O(nlist * d * log nlist) host work; larger indexes keep the list-id order
        std::vector<float> hc((size_t)h->nlist * h->d);
        HIP_TRY(hipMemcpy(hc.data(), h->coarse.p, bytes, hipMemcpyDeviceToHost));
        TRY(build_screen(h, hc.data(), h->coarse.as<float>(), h->nlist, h->d, h->screen));
        std::vector<int> rank;
        spatial_list_rank(hc.data(), h->nlist, h->d, rank);
        TRY(h->list_rank.reserve((size_t)h->nlist * sizeof(int)));
        HIP_TRY(hipMemcpy(h->list_rank.p, rank.data(), (size_t)h->nlist * sizeof(int), hipMemcpyHostToDevice));
        // 8 partitions of neighbouring lists (equal list counts along the spatial order), one per XCD
        std::vector<uint8_t> part((size_t)h->nlist);
        for (int i = 0; i < h->nlist; i++) part[(size_t)i] = (uint8_t)(((int64_t)rank[(size_t)i] * 8) / h->nlist);
        TRY(h->list_part.reserve((size_t)h->nlist));
        HIP_TRY(hipMemcpy(h->list_part.p, part.data(), (size_t)h->nlist, hipMemcpyHostToDevice));
        h->have_rank = true;
    }
    h->have_coarse = true;
    h->imi_nbits = 0;
    h->term2_valid = false;
    h->term2h_valid = false;
    h->coarse_s_stride = 0;          // the sampled tiles belong to the old centroids
    return VLQ_OK;
}

int vlq_ivfpq_set_imi_centroids(vlq_ivfpq_t h, int imi_nbits, const float* centroids) {
    if (!h || !centroids) return fail(VLQ_ERR_INVALID, "null argument");
    if (imi_nbits < 1 || imi_nbits > 15) return fail(VLQ_ERR_INVALID, "imi_nbits=%d outside 1..15", imi_nbits);
    if ((int64_t)h->nlist != (int64_t(1) << (2 * imi_nbits)))
        return fail(VLQ_ERR_INVALID, "nlist=%d must be 4^imi_nbits for a 2 x %d-bit multi-index", h->nlist, imi_nbits);
    if (h->d % 2 != 0 || h->M % 2 != 0)   // IndexIVFPQ.cpp:404: pq.M % miq->pq.M == 0
        return fail(VLQ_ERR_INVALID, "d and M must be even for a 2-way multi-index");
    TRY(set_dev(h));
    const int64_t kc = int64_t(1) << imi_nbits;
    const int dc = h->d / 2;
    const size_t bytes = (size_t)2 * kc * dc * sizeof(float);
    TRY(h->imi_cent.reserve(bytes));
    TRY(h->imi_norm.reserve((size_t)2 * kc * sizeof(float)));
    TRY(h->imi_virtual.reserve((size_t)kc * h->d * sizeof(float)));
    std::vector<float> hc((size_t)2 * kc * dc), hv((size_t)kc * h->d);
    HIP_TRY(hipMemcpy(hc.data(), centroids, bytes, hipMemcpyDefault));
    for (int64_t i = 0; i < kc; i++)      // IndexIVFPQ.cpp:440-448
        for (int m = 0; m < 2; m++)
            memcpy(&hv[(size_t)i * h->d + m * dc], &hc[((size_t)m * kc + i) * dc], sizeof(float) * dc);
    HIP_TRY(hipMemcpyAsync(h->imi_cent.p, hc.data(), bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->imi_virtual.p, hv.data(), hv.size() * 4, hipMemcpyHostToDevice, h->stream));
    vlq::launch_row_norms(h->imi_cent.as<float>(), 2 * kc, dc, h->imi_norm.as<float>(), h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int m = 0; m < 2; m++)      // float16 screen of each half's distance table (coarse_screen.hip)
        TRY(build_screen(h, hc.data() + (size_t)m * kc * dc, h->imi_cent.as<float>() + (size_t)m * kc * dc, (int)kc, dc, h->imi_screen[m]));
    h->imi_nbits = imi_nbits;
    h->have_coarse = true;
    h->term2_valid = false;
    h->term2h_valid = false;
    return VLQ_OK;
}

int vlq_ivfpq_set_pq_centroids(vlq_ivfpq_t h, const float* centroids) {
    if (!h || !centroids) return fail(VLQ_ERR_INVALID, "null argument");
    TRY(set_dev(h));
    const size_t n = (size_t)h->M * h->ksub;
    const size_t bytes = n * h->dsub * sizeof(float);
    TRY(h->pq.reserve(bytes));
    TRY(h->rnorm.reserve(n * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(h->pq.p, centroids, bytes, hipMemcpyDefault, h->stream));
    // r_norms (IndexIVFPQ.cpp:411-416)
    vlq::launch_row_norms(h->pq.as<float>(), (int64_t)n, h->dsub, h->rnorm.as<float>(), h->stream);
    TRY(h->pq_t.reserve(bytes));
    vlq::launch_transpose_pq(h->pq.as<float>(), h->M, h->ksub, h->dsub, h->pq_t.as<float>(), h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->have_pq = true;
    h->term2_valid = false;
    h->term2h_valid = false;
    return VLQ_OK;
}

int vlq_ivfpq_set_search_options(vlq_ivfpq_t h, int by_residual, int use_precomputed_table,
                                 int64_t max_codes) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (use_precomputed_table != 0 && use_precomputed_table != 1)
        return fail(VLQ_ERR_UNSUPPORTED, "use_precomputed_table=%d (only 0 and 1; 2 = IMI not built)",
                    use_precomputed_table);
    if (max_codes < 0) return fail(VLQ_ERR_INVALID, "max_codes < 0");
    h->by_residual = by_residual ? 1 : 0;
    h->use_precomputed_table = use_precomputed_table;
    h->max_codes = max_codes;
    return VLQ_OK;
}

int vlq_ivfpq_set_float16_tables(vlq_ivfpq_t h, int enable) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (enable && !(h->M == 16 && h->ksub == 256))
        return fail(VLQ_ERR_UNSUPPORTED, "float16 look-up tables are built for 16 x 8-bit codes only");
    h->fp16_tables = enable != 0;
    return VLQ_OK;
}

int vlq_ivfpq_set_coarse_screen(vlq_ivfpq_t h, int mode) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (mode != 0 && mode != 1) return fail(VLQ_ERR_INVALID, "coarse screen mode %d (0 = off, 1 = on)", mode);
    h->coarse_screen = mode;
    if (mode) { h->screen_rows_seen = h->screen_rows_copied = 0; if (h->screen_cnt_host) *h->screen_cnt_host = 0; }
    if (mode && h->ws_screen_cnt.p) { TRY(set_dev(h)); HIP_TRY(hipMemsetAsync(h->ws_screen_cnt.p, 0, 8, h->stream)); }
    return VLQ_OK;
}

int vlq_ivfpq_coarse_screen_state(vlq_ivfpq_t h, int* enabled, uint64_t* rows, uint32_t* undecided) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (h->screen_cnt_host) { TRY(set_dev(h)); HIP_TRY(hipStreamSynchronize(h->stream)); }     // the mirror is up to date after this
    if (h->coarse_screen && h->screen_cnt_host && h->screen_rows_copied >= 1024 &&
        (uint64_t)*h->screen_cnt_host * 200 > h->screen_rows_copied)
        h->coarse_screen = 0;
    if (enabled) *enabled = (h->coarse_screen && (h->imi_nbits > 0 ? (h->imi_screen[0].ok && h->imi_screen[1].ok) : h->screen.ok)) ? 1 : 0;
    if (rows) *rows = h->screen_rows_seen;
    if (undecided) *undecided = h->screen_cnt_host ? *h->screen_cnt_host : 0u;
    return VLQ_OK;
}

int vlq_ivfpq_set_scan_schedule(vlq_ivfpq_t h, int mode) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (mode < 0 || mode > 4) return fail(VLQ_ERR_INVALID, "scan schedule %d outside 0..4", mode);
    h->scan_schedule = mode;
    return VLQ_OK;
}

int vlq_ivfpq_set_lists(vlq_ivfpq_t h, const uint8_t* codes, const int64_t* ids,
                        const int64_t* list_offsets) {
    if (!h || !list_offsets) return fail(VLQ_ERR_INVALID, "null argument");
    TRY(set_dev(h));
    std::vector<int64_t> off((size_t)h->nlist + 1);
    HIP_TRY(hipMemcpy(off.data(), list_offsets, off.size() * 8, hipMemcpyDefault));
    if (off[0] != 0) return fail(VLQ_ERR_INVALID, "list_offsets[0] != 0");
    for (int i = 0; i < h->nlist; i++) {
        if (off[i + 1] < off[i]) return fail(VLQ_ERR_INVALID, "list_offsets not monotone at %d", i);
        if (off[i + 1] - off[i] >= (int64_t(1) << 31))
            return fail(VLQ_ERR_UNSUPPORTED, "list %d longer than 2^31", i);
    }
    const int64_t ntotal = off[h->nlist];
    if (ntotal > 0 && (!codes || !ids)) return fail(VLQ_ERR_INVALID, "null codes/ids");
    TRY(h->codes.reserve((size_t)ntotal * h->M + 16));
    TRY(h->ids.reserve((size_t)ntotal * 8 + 16));
    if (ntotal > 0) {
        HIP_TRY(hipMemcpyAsync(h->codes.p, codes, (size_t)ntotal * h->M, hipMemcpyDefault, h->stream));
        HIP_TRY(hipMemcpyAsync(h->ids.p, ids, (size_t)ntotal * 8, hipMemcpyDefault, h->stream));
    }
    std::vector<int64_t> len((size_t)h->nlist);
    for (int i = 0; i < h->nlist; i++) len[(size_t)i] = off[(size_t)i + 1] - off[(size_t)i];   // packed: capacity == length
    HIP_TRY(hipMemcpyAsync(h->list_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->list_len.p, len.data(), len.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->h_list_off = off;
    h->h_list_len.swap(len);
    h->h_lists_stale = false;
    h->ntotal = ntotal;
    h->have_lists = true;
    // one-time costs of the search path that depend on the trained state only -- the precomputed table (the reference builds
    // it at train / read_index time: IndexIVFPQ::precompute_table) and the code objects of the search kernels -- are paid here,
    // with the lists, not inside the first search a caller may be timing (bench.py first_call_ms: 2.7 -> see profiles/)
    if (h->have_coarse && h->have_pq) {
        TRY(ensure_term2(h));
        vlq::preload_search_kernels();
    }
    return VLQ_OK;
}

int64_t vlq_ivfpq_ntotal(vlq_ivfpq_t h) { return h ? h->ntotal : -1; }

int vlq_ivfpq_list_length(vlq_ivfpq_t h, int list_id, int64_t* len) {
    if (!h || !len) return fail(VLQ_ERR_INVALID, "null argument");
    if (list_id < 0 || list_id >= h->nlist) return fail(VLQ_ERR_INVALID, "list id out of range");
    if (h->h_lists_stale) {
        TRY(set_dev(h));
        vlq::ListStore ls = list_store(h);
        TRY(vlq::lists_sync_host(ls, h->stream));
    }
    *len = h->h_list_len[list_id];
    return VLQ_OK;
}

int vlq_ivfpq_get_list(vlq_ivfpq_t h, int list_id, uint8_t* codes_out, int64_t* ids_out) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (list_id < 0 || list_id >= h->nlist) return fail(VLQ_ERR_INVALID, "list id out of range");
    TRY(set_dev(h));
    {
        vlq::ListStore ls = list_store(h);
        TRY(vlq::lists_sync_host(ls, h->stream));
    }
    const int64_t o = h->h_list_off[list_id], len = h->h_list_len[list_id];
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (len > 0 && codes_out)
        HIP_TRY(hipMemcpy(codes_out, h->codes.as<uint8_t>() + o * h->M, (size_t)len * h->M, hipMemcpyDeviceToHost));
    if (len > 0 && ids_out)
        HIP_TRY(hipMemcpy(ids_out, h->ids.as<int64_t>() + o, (size_t)len * 8, hipMemcpyDeviceToHost));
    return VLQ_OK;
}

int vlq_ivfpq_coarse_search(vlq_ivfpq_t h, int64_t n, const float* x, int nprobe,
                            float* coarse_dis, int64_t* keys) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (!h->have_coarse) return fail(VLQ_ERR_STATE, "coarse centroids not set (index not trained)");
    TRY(check_search_args(h, n, x, nprobe, 1, coarse_dis, keys));
    if (n == 0) return VLQ_OK;
    TRY(set_dev(h));
    const void* xd;
    TRY(stage_in(h, x, (size_t)n * h->d * 4, h->ws_x, &xd));
    void *cd, *kd;
    bool copy_c, copy_k;
    TRY(stage_out(coarse_dis, (size_t)n * nprobe * 4, h->ws_cdis, &cd, &copy_c));
    TRY(stage_out(keys, (size_t)n * nprobe * 8, h->ws_keys, &kd, &copy_k));
    TRY(coarse_dev(h, n, (const float*)xd, nprobe, (float*)cd, (int64_t*)kd));
    return finish_outputs(h, copy_c, coarse_dis, cd, (size_t)n * nprobe * 4, copy_k, keys, kd,
                          (size_t)n * nprobe * 8);
}

// More probes than one scan launch takes (the CPU class has no limit: tests/sift1b_imi_pq.cpp asks for 2048): the probe list
// is cut into runs of <= 1024 in coarse order (strided device copies), every run is scanned, and the rows are joined by
// (distance, run, place in the run's row) -- the (distance, scan position) order of one long scan (merge_topk_kernel: ties go
// to the lower part, then the lower rank).  Pages of 32 768 queries bound the run buffers.
static int scan_runs_dev(vlq_ivfpq_t h, int64_t n, const float* xd, const int64_t* kd, const float* cd, int nprobe, int k, float* Dd,
                         int64_t* Id, int store_pairs) {
    if (nprobe <= VLQ_MAX_NPROBE) return scan_dev(h, n, xd, kd, cd, nprobe, k, Dd, Id, store_pairs);
    if (h->max_codes != 0)
        return fail(VLQ_ERR_UNSUPPORTED, "max_codes=%lld with nprobe=%d > %d: the limit would apply to every run of probes, not to the "
                    "whole list (IndexIVFPQ.cpp:1052)", (long long)h->max_codes, nprobe, VLQ_MAX_NPROBE);
    const int nruns = (nprobe + VLQ_MAX_NPROBE - 1) / VLQ_MAX_NPROBE;
    const int64_t page = 32768;
    const int64_t np = std::min(n, page);
    TRY(h->ws_keys_run.reserve((size_t)np * VLQ_MAX_NPROBE * 8));
    TRY(h->ws_cdis_run.reserve((size_t)np * VLQ_MAX_NPROBE * 4));
    TRY(h->ws_Dr.reserve((size_t)nruns * np * k * 4));
    TRY(h->ws_Ir.reserve((size_t)nruns * np * k * 8));
    const uint64_t nq0 = h->stat_nq;
    for (int64_t i0 = 0; i0 < n; i0 += page) {
        const int64_t ni = std::min(page, n - i0);
        for (int r = 0; r < nruns; r++) {
            const int p0 = r * VLQ_MAX_NPROBE, pn = std::min(VLQ_MAX_NPROBE, nprobe - p0);
            HIP_TRY(hipMemcpy2DAsync(h->ws_keys_run.p, (size_t)pn * 8, kd + i0 * nprobe + p0, (size_t)nprobe * 8, (size_t)pn * 8, (size_t)ni,
                                     hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipMemcpy2DAsync(h->ws_cdis_run.p, (size_t)pn * 4, cd + i0 * nprobe + p0, (size_t)nprobe * 4, (size_t)pn * 4, (size_t)ni,
                                     hipMemcpyDeviceToDevice, h->stream));
            TRY(scan_dev(h, ni, xd + i0 * h->d, h->ws_keys_run.as<int64_t>(), h->ws_cdis_run.as<float>(), pn, k,
                         h->ws_Dr.as<float>() + (size_t)r * ni * k, h->ws_Ir.as<int64_t>() + (size_t)r * ni * k, store_pairs));
        }
        vlq::launch_merge_topk(h->ws_Dr.as<float>(), h->ws_Ir.as<int64_t>(), ni, k, nruns, Dd + i0 * k, Id + i0 * k, h->stream);
        HIP_TRY(hipGetLastError());
    }
    h->stat_nq = nq0 + (uint64_t)n;          // (the reference counts a query once, however its probes were cut)
    return VLQ_OK;
}

int vlq_ivfpq_search_preassigned(vlq_ivfpq_t h, int64_t n, const float* x, const int64_t* keys,
                                 const float* coarse_dis, int nprobe, int k, float* D, int64_t* I,
                                 int store_pairs) {
    TRY(check_ready(h, true));
    // more probes than one scan takes: the CPU class has no limit (tests/sift1b_imi_pq.cpp asks for 2048) -- see below
    TRY(check_search_args(h, n, x, std::min(nprobe, VLQ_MAX_NPROBE), k, D, I));
    if (nprobe > 64 * VLQ_MAX_NPROBE) return fail(VLQ_ERR_INVALID, "nprobe=%d beyond %d", nprobe, 64 * VLQ_MAX_NPROBE);
    if (n > 0 && (!keys || !coarse_dis)) return fail(VLQ_ERR_INVALID, "null keys/coarse_dis");
    if (n == 0) return VLQ_OK;
    TRY(set_dev(h));
    h->order_hist_ready = false;              // (the caller's keys: no histogram came with them)
    const void *xd, *kd, *cd;
    TRY(stage_in(h, x, (size_t)n * h->d * 4, h->ws_x, &xd));
    TRY(stage_in(h, keys, (size_t)n * nprobe * 8, h->ws_keys_in, &kd));
    TRY(stage_in(h, coarse_dis, (size_t)n * nprobe * 4, h->ws_cdis_in, &cd));
    void *Dd, *Id;
    bool copyD, copyI;
    TRY(stage_out(D, (size_t)n * k * 4, h->ws_D, &Dd, &copyD));
    TRY(stage_out(I, (size_t)n * k * 8, h->ws_I, &Id, &copyI));
    TRY(scan_runs_dev(h, n, (const float*)xd, (const int64_t*)kd, (const float*)cd, nprobe, k, (float*)Dd, (int64_t*)Id, store_pairs));
    TRY(finish_outputs(h, copyD, D, Dd, (size_t)n * k * 4, copyI, I, Id, (size_t)n * k * 8));
    // host outputs: the call has synchronised, so an invalid key is reported here and now; device
    // outputs: the call stays asynchronous and the flag surfaces at the next vlq_ivfpq_stats()
    if (copyD || copyI) TRY(read_bad_key(h));
    return VLQ_OK;
}

// Host buffers (the reference drivers' calling convention): copy in, search, copy out on the index's
// stream.  Two overlapped variants were built and measured on the bench batch (10 000 queries, pageable
// numpy buffers, tools/host_buffers.py) and LOST to this plain sequence (1.09 ms = 1.25x the
// device-resident step): three pages of 10/30/60 % with the copies of one page beside the search of
// another 1.21 ms (1.39x: three small searches cost more than the 0.11 + 0.04 ms of copies they hide);
// chunked copy-in with the coarse stage chunk by chunk behind it and ONE scan 1.13 ms (1.32x).  A
// pageable hipMemcpyAsync blocks the host, so nothing can be enqueued behind it without pinning the
// caller's pages; DESIGN.md section 7.
// Round 3, page-locked buffers (GpuResources::getPinnedMemory): result rows are written by the scan kernel straight
// into the caller's memory (free: 0.848 ms against 0.848 device-resident; the D2H copies cost 0.055 ms), the 5 MB of
// queries cost their 0.105 ms at 51 GB/s.  Copy-in on a second stream was tried twice more with page-locked sources --
// two halves, each with its own coarse call (0.986 ms against 0.974 unsplit: the copy of the second half does run
// beside the first GEMM, rocprofv3 --memory-copy-trace, but two half-size GEMM + select pairs cost 190 us against
// 161 and the event wait 15 us); four chunks beside four partial GEMMs of ONE matrix, one select (1.000 ms) -- and
// removed: cross-stream event waits cost more than the 50-75 us of copy they hide.  (Polling hipStreamQuery instead of
// hipStreamSynchronize at the end: 0.977 against 0.984 ms, noise.)
int vlq_ivfpq_search(vlq_ivfpq_t h, int64_t n, const float* x, int nprobe, int k, float* D,
                     int64_t* I) {
    TRY(check_ready(h, true));
    TRY(check_search_args(h, n, x, nprobe, k, D, I));
    if (n == 0) return VLQ_OK;
    TRY(set_dev(h));
    TRY(h->ws_keys.reserve((size_t)n * nprobe * 8));
    TRY(h->ws_cdis.reserve((size_t)n * nprobe * 4));
    void *Dd, *Id;
    bool copyD, copyI, zcD, zcI;
    TRY(stage_out(D, (size_t)n * k * 4, h->ws_D, &Dd, &copyD, &zcD));
    TRY(stage_out(I, (size_t)n * k * 8, h->ws_I, &Id, &copyI, &zcI));
    const void* xd = nullptr;
    TRY(stage_in(h, x, (size_t)n * h->d * 4, h->ws_x, &xd));
    // the scan order's histogram rides on the coarse stage's last kernel when one coarse page and one scan page serve the batch
    // (kernels.h OrderHist; the single-workgroup ordering of small batches does not use it)
    h->order_hist = vlq::OrderHist();
    h->order_hist_ready = false;
    if (h->imi_nbits == 0 && n > 2048 && n <= 32768 && n <= query_page(h) && h->nlist <= (1 << 22) && !getenv("VLQ_ORDER_HIST_OFF")) {
        const size_t stride = vlq::query_order_bins_padded(h->nlist);
        TRY(h->ws_hist.reserve(2 * stride * sizeof(int)));
        HIP_TRY(hipMemsetAsync(h->ws_hist.p, 0, 2 * stride * sizeof(int), h->stream));
        h->order_hist.hist = h->ws_hist.as<int>();
        h->order_hist.list_rank = h->have_rank ? h->list_rank.as<int>() : nullptr;
        h->order_hist.nlist = h->nlist;
        vlq::query_order_bins(h->nlist, &h->order_hist.shift, &h->order_hist.nbins);
    }
    // IndexIVFPQ::search (IndexIVFPQ.cpp:1063-1081): quantizer->search, then search_knn_with_key
    TRY(coarse_dev(h, n, (const float*)xd, nprobe, h->ws_cdis.as<float>(), h->ws_keys.as<int64_t>()));
    h->order_hist.hist = nullptr;            // (only this call's coarse stage may add to the counts)
    TRY(scan_runs_dev(h, n, (const float*)xd, h->ws_keys.as<int64_t>(), h->ws_cdis.as<float>(), nprobe, k,
                      (float*)Dd, (int64_t*)Id, 0));
    TRY(finish_outputs(h, copyD, D, Dd, (size_t)n * k * 4, copyI, I, Id, (size_t)n * k * 8));
    if ((zcD || zcI) && !(copyD || copyI)) HIP_TRY(hipStreamSynchronize(h->stream));    // rows in the caller's memory on return
    return VLQ_OK;
}

int vlq_ivfpq_query_tables(vlq_ivfpq_t h, int64_t n, const float* x, int inner_product, float* out) {
    TRY(check_ready(h, false));
    if (n < 0 || (n > 0 && (!x || !out))) return fail(VLQ_ERR_INVALID, "bad argument");
    if (n == 0) return VLQ_OK;
    TRY(set_dev(h));
    const size_t E = (size_t)h->M * h->ksub;
    const void* xd;
    TRY(stage_in(h, x, (size_t)n * h->d * 4, h->ws_x, &xd));
    TRY(h->ws_qtab.reserve((size_t)n * E * 4));
    vlq::launch_pq_tables((const float*)xd, n, h->d, h->pq.as<float>(), h->M, h->ksub, h->dsub, nullptr,
                          inner_product ? 0 : 1, h->ws_qtab.as<float>(), h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->ws_qtab.p, (size_t)n * E * 4, hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VLQ_OK;
}

int vlq_ivfpq_get_precomputed_table(vlq_ivfpq_t h, float* out) {
    TRY(check_ready(h, false));
    if (!out) return fail(VLQ_ERR_INVALID, "null out");
    if (!(h->by_residual && h->use_precomputed_table == 1))
        return fail(VLQ_ERR_STATE, "precomputed table not in use");
    TRY(set_dev(h));
    TRY(ensure_term2(h));
    const size_t rows = h->imi_nbits > 0 ? (size_t(1) << h->imi_nbits) : (size_t)h->nlist;
    HIP_TRY(hipMemcpyAsync(out, h->term2.p, rows * h->M * h->ksub * 4, hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return VLQ_OK;
}

int vlq_ivfpq_stats(vlq_ivfpq_t h, uint64_t* nq, uint64_t* ncode, int reset) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    TRY(set_dev(h));
    unsigned long long st[64] = {0};
    HIP_TRY(hipMemcpyAsync(st, h->stats.p, 512, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (nq) *nq = h->stat_nq;
    if (ncode) *ncode = st[0];
    if (st[8 + 14] && getenv("VLQ_SCAN16_PHASES")) {    // only a scan16 built with -DVLQ_SCAN16_PHASES writes these (scan16.hip)
        static const char* names[14] = {"set-up: second barrier, first prefetch", "barrier before the table build", "wait for the prefetched row (vmcnt)",
                                        "table build + next prefetch issued", "barrier after the build", "admission bound refresh",
                                        "gather trips + selection", "merge + rows out", "set-up: placement", "set-up: probe keys, list offsets",
                                        "set-up: per-query table", "set-up: first barrier", "set-up: prefix sums, live probes (wave 0)",
                                        "set-up: walking order (wave 0)"};
        static const int order[14] = {8, 9, 10, 11, 12, 13, 0, 1, 2, 3, 4, 5, 6, 7};
        for (int w = 0; w < 2; w++) {
            const unsigned long long* o = st + 8 + 20 * w;
            if (!o[14]) continue;
            const double wg = (double)o[14], probes = (double)(o[15] & 0xffffffffull), trips = (double)(o[15] >> 32);
            fprintf(stderr, "[scan16 phases] wave %d: %llu workgroups sampled, %.1f probes and %.1f trips each, %.0f cycles = %.2f us per workgroup "
                            "(clock %.2f GHz)\n", w, o[14], probes / wg, trips / wg, o[16] / wg, o[17] * 0.01 / wg, o[16] / (o[17] * 10.0));
            fprintf(stderr, "[scan16 phases]   (of the gather trips: the 16 gathers + adds of a trip %.0f cycles per trip, %.0f per workgroup)\n",
                    o[18] / trips, o[18] / wg);
            for (int j = 0; j < 14; j++) {
                const int i = order[j];
                fprintf(stderr, "[scan16 phases]   %-44s %9.0f cycles per workgroup  %7.1f per probe  %5.1f %%\n", names[i], o[i] / wg,
                        o[i] / probes, 100.0 * o[i] / o[16]);
            }
        }
    }
    if (st[5] && getenv("VLQ_PHASE_TIMING"))     // only kernels built with -DVLQ_PHASE_TIMING write these
        fprintf(stderr, "[phase timing] per workgroup: prologue %.2f us, loop %.2f us, tail %.2f us (%llu workgroups)\n",
                st[2] * 0.01 / st[5], st[3] * 0.01 / st[5], st[4] * 0.01 / st[5], st[5]);
    const int bad = (int)(st[1] & 0xffffffffu);
    if (reset) {
        HIP_TRY(hipMemsetAsync(h->stats.p, 0, 512, h->stream));
        h->stat_nq = 0;
    } else if (bad) {       // the flag is consumed by the error it raises; the counters stay
        HIP_TRY(hipMemsetAsync(reinterpret_cast<char*>(h->stats.p) + 8, 0, 8, h->stream));
    }
    // the reference aborts the search on an out-of-range key (IndexIVFPQ.cpp:1008-1011)
    if (bad) return bad_flag_error(bad);
    return VLQ_OK;
}

int vlq_ivfpq_last_scan_info(vlq_ivfpq_t h, char* buf, int cap) {
    if (!h || !buf || cap < 1) return fail(VLQ_ERR_INVALID, "null argument");
    TRY(set_dev(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const char* order = "coarse-distance order";
    int shared = -1;
    if (h->last_walk_first >= 0) {
        order = "list-id walk";
        if (h->last_walk_counts) {      // decided on the device from walk_stat_kernel's counts (the handle's own copy of them)
            int v[32];
            HIP_TRY(hipMemcpy(v, h->walk_counts.p, sizeof(v), hipMemcpyDeviceToHost));
            shared = 0;
            for (int x : v) shared += x;
            if (shared > h->last_walk_limit) order = "coarse-distance order";
        }
    }
    int period = 0, launch_period = 0;      // XCD 0: the running mean of the measured walk times / what the last launch ran with
    if (h->walk_state.p) {
        int ws[8 * 16];
        HIP_TRY(hipMemcpy(ws, h->walk_state.p, sizeof(ws), hipMemcpyDeviceToHost));
        period = ws[0]; launch_period = ws[1];
    }
    snprintf(buf, (size_t)cap, "kernel=%s order=%s first=%d shared=%d/%d limit=%d period_ticks=%d launch_period_ticks=%d",
             h->last_scan[0] ? h->last_scan : "none", order, h->last_walk_first, shared, h->last_walk_samples, h->last_walk_limit, period,
             launch_period);
    return VLQ_OK;
}

int vlq_ivfpq_reset_walk_state(vlq_ivfpq_t h) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    TRY(set_dev(h));
    if (h->walk_state.p) HIP_TRY(hipMemsetAsync(h->walk_state.p, 0, 8 * 16 * sizeof(int), h->stream));
    h->walk_stat_calls = 0;
    return VLQ_OK;
}

int vlq_ivfpq_profile(vlq_ivfpq_t h, int enable) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    h->prof = enable != 0;
    h->prof_scan_only = enable == 2 || enable == 3;
    h->prof_every = enable == 3 ? 4 : 1;
    h->prof_seq = 0;
    return VLQ_OK;
}

int vlq_ivfpq_profile_read(vlq_ivfpq_t h, double ms[3], int64_t* calls, int reset) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    TRY(set_dev(h));
    drain_profile(h);
    if (ms) { ms[0] = h->prof_ms[0]; ms[1] = h->prof_ms[1]; ms[2] = h->prof_ms[2]; }
    if (calls) *calls = h->prof_calls;
    if (reset) { h->prof_ms[0] = h->prof_ms[1] = h->prof_ms[2] = 0; h->prof_calls = 0; }
    return VLQ_OK;
}

int vlq_merge_topk(int device, void* hip_stream, int64_t nq, int k, int nparts, const float* D_parts,
                   const int64_t* I_parts, float* D, int64_t* I) {
    if (nq < 0 || k < 1 || k > VLQ_MAX_K || nparts < 1) return fail(VLQ_ERR_INVALID, "bad argument");
    if (nq == 0) return VLQ_OK;
    if (!D_parts || !I_parts || !D || !I) return fail(VLQ_ERR_INVALID, "null buffer");
    if ((int64_t)nparts * k >= (int64_t(1) << 31)) return fail(VLQ_ERR_UNSUPPORTED, "nparts*k too large");
    if (!is_device_ptr(D_parts) || !is_device_ptr(I_parts) || !is_device_ptr(D) || !is_device_ptr(I))
        return fail(VLQ_ERR_INVALID, "vlq_merge_topk takes device buffers (the per-shard results were all-gathered on the device)");
    HIP_TRY(hipSetDevice(device));
    vlq::launch_merge_topk(D_parts, I_parts, nq, k, nparts, D, I, reinterpret_cast<hipStream_t>(hip_stream));
    HIP_TRY(hipGetLastError());
    return VLQ_OK;
}

// device pointers in and out: list assignment (quantizer->assign, IndexIVFPQ.cpp:205 = 1-NN
// search) and PQ codes of the residuals
static int encode_dev(vlq_ivfpq_t h, int64_t n, const float* xd, int64_t* ad, uint8_t* cd) {
    TRY(h->ws_misc.reserve((size_t)n * 4));
    TRY(coarse_dev(h, n, xd, 1, h->ws_misc.as<float>(), ad));
    vlq::launch_residual_encode(xd, n, h->d, h->imi_nbits > 0 ? h->imi_cent.as<float>() : h->coarse.as<float>(),
                                ad, h->by_residual, h->pq.as<float>(), h->M, h->ksub, h->dsub, cd, h->stream,
                                h->imi_nbits);
    HIP_TRY(hipGetLastError());
    return VLQ_OK;
}

int vlq_ivfpq_encode(vlq_ivfpq_t h, int64_t n, const float* x, int64_t* assign, uint8_t* codes) {
    TRY(check_ready(h, false));
    if (n < 0 || (n > 0 && (!x || !assign || !codes))) return fail(VLQ_ERR_INVALID, "bad argument");
    if (n == 0) return VLQ_OK;
    TRY(set_dev(h));
    const void* xd;
    TRY(stage_in(h, x, (size_t)n * h->d * 4, h->ws_x, &xd));
    void *ad, *cd;
    bool copy_a, copy_c;
    TRY(stage_out(assign, (size_t)n * 8, h->ws_assign, &ad, &copy_a));
    TRY(stage_out(codes, (size_t)n * h->M, h->ws_codes, &cd, &copy_c));
    TRY(encode_dev(h, n, (const float*)xd, (int64_t*)ad, (uint8_t*)cd));
    return finish_outputs(h, copy_a, assign, ad, (size_t)n * 8, copy_c, codes, cd, (size_t)n * h->M);
}

int vlq_ivfpq_encode_preassigned(vlq_ivfpq_t h, int64_t n, const float* x, const int64_t* assign, uint8_t* codes) {
    TRY(check_ready(h, false));
    if (n < 0 || (n > 0 && (!x || !assign || !codes))) return fail(VLQ_ERR_INVALID, "bad argument");
    if (n == 0) return VLQ_OK;
    TRY(set_dev(h));
    const void *xd, *ad;
    TRY(stage_in(h, x, (size_t)n * h->d * 4, h->ws_x, &xd));
    TRY(stage_in(h, assign, (size_t)n * 8, h->ws_assign, &ad));
    void* cd;
    bool copy_c;
    TRY(stage_out(codes, (size_t)n * h->M, h->ws_codes, &cd, &copy_c));
    vlq::launch_residual_encode((const float*)xd, n, h->d, h->imi_nbits > 0 ? h->imi_cent.as<float>() : h->coarse.as<float>(),
                                (const int64_t*)ad, h->by_residual, h->pq.as<float>(), h->M, h->ksub, h->dsub, (uint8_t*)cd,
                                h->stream, h->imi_nbits);
    HIP_TRY(hipGetLastError());
    return finish_outputs(h, false, nullptr, nullptr, 0, copy_c, codes, cd, (size_t)n * h->M);
}

int vlq_ivfpq_add(vlq_ivfpq_t h, int64_t n, const float* x, const int64_t* xids) {
    TRY(check_ready(h, false));
    if (n < 0 || (n > 0 && !x)) return fail(VLQ_ERR_INVALID, "bad argument");
    if (n == 0) return VLQ_OK;
    TRY(set_dev(h));
    // encode and append on the device (IndexIVFPQ.cpp:192-272; gpu/impl/IVFPQ.cu:197-426,
    // gpu/impl/InvertedListAppend.cu:122-247): nothing but an overflow flag and two totals visits the host
    const void* xd;
    TRY(stage_in(h, x, (size_t)n * h->d * 4, h->ws_x, &xd));
    TRY(h->ws_assign.reserve((size_t)n * 8));
    TRY(h->ws_codes.reserve((size_t)n * h->M));
    TRY(encode_dev(h, n, (const float*)xd, h->ws_assign.as<int64_t>(), h->ws_codes.as<uint8_t>()));
    const void* idd = nullptr;
    if (xids) TRY(stage_in(h, xids, (size_t)n * 8, h->ws_keys_in, &idd));
    vlq::ListStore ls = list_store(h);
    TRY(vlq::lists_append(ls, h->ws_append, n, h->ws_assign.as<int64_t>(), nullptr, h->ws_codes.as<uint8_t>(),
                          nullptr, (const int64_t*)idd, h->ntotal, h->stream));
    h->ntotal += n;                                             // IndexIVFPQ.cpp:271
    // (as in vlq_ivfpq_set_lists: the precomputed table and the search kernels' code objects belong to building the index)
    if (h->have_coarse && h->have_pq) {
        TRY(ensure_term2(h));
        vlq::preload_search_kernels();
    }
    return VLQ_OK;
}



int vlq_ivfpq_reserve_memory(vlq_ivfpq_t h, int64_t num_vecs) {
    if (!h || num_vecs < 0) return fail(VLQ_ERR_INVALID, "bad argument");
    TRY(set_dev(h));
    vlq::ListStore ls = list_store(h);
    return vlq::lists_reserve(ls, num_vecs, h->stream);
}

int vlq_ivfpq_reclaim_memory(vlq_ivfpq_t h, uint64_t* bytes_reclaimed) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    TRY(set_dev(h));
    vlq::ListStore ls = list_store(h);
    return vlq::lists_reclaim(ls, bytes_reclaimed, h->stream);
}

}  // extern "C"
