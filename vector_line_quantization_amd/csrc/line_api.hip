// C ABI of the VLQ index (include/vlq_line.h).  Host orchestration only; the coarse
// stage, the term2 table and the per-query tables are the IVFPQ library's kernels
// (an embedded vlq_ivfpq handle), the line-specific stages are line.hip.
#include "../../include/vlq_line.h"
#include "handle.h"
#include "lists.h"
#include "line.h"

struct vlq_line_s {
    vlq_ivfpq_t base = nullptr;          // coarse centroids, PQ, term2, workspace, stream
    int nedge = 0, nlambda = 0;
    int64_t nlines = 0, ntotal = 0, ntotal_added = 0;
    DevBuf edge_info, edge_dist, lambda_info, codes, lambdas, ids, line_off, line_len;   // lists.h layout
    bool have_graph = false, have_lambda = false;
    // float16 look-up tables (GpuIndexIVFPQConfig::useFloat16LookupTables): half(term2), built on demand
    bool fp16_tables = false, term2h_valid = false;
    // scan-kernel timing (vlq_line_profile): event pairs recorded around the scan launches, drained on read
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms = 0;
    int64_t prof_calls = 0;
    int scan_parts = 0;                  // vlq_line_set_scan_parts: workgroups per query of the line16c scan, 0 = automatic
    int row_mode = 0;                    // vlq_line_set_row_mode: 0 auto, 1 stored term-2 rows, 2 rebuilt rows
    DevBuf term2h, ws_qtabh;
    // per-code constants of the stored codes (line16c.hip), one array per table precision; rebuilt lazily after anything
    // they depend on changed (codes / lambda bytes / line layout, term 2 = coarse + PQ centroids, graph, lambda codebook)
    DevBuf pconst, pconsth, ws_part_keys;
    bool pconst_valid = false, pconsth_valid = false;
    std::vector<int64_t> h_line_off, h_line_len;
    bool h_lines_stale = false;
    AppendWs ws_append;
    std::vector<float> h_lambda;
    DevBuf ws_near, ws_line, ws_lamf, ws_lamb, ws_res, ws_codes, ws_sel_line, ws_sel_b2, ws_sel_g, ws_sel_meta, ws_sel_cnt,
        ws_x, ws_D, ws_I, ws_keys, ws_cdis, stats;
};

namespace {

int need(vlq_line_t h, bool graph, bool lambda, bool pq) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (!h->base->have_coarse) return fail(VLQ_ERR_STATE, "coarse centroids not set");
    if (graph && !h->have_graph) return fail(VLQ_ERR_STATE, "centroid graph not set/built");
    if (lambda && !h->have_lambda) return fail(VLQ_ERR_STATE, "lambda codebook not set");
    if (pq && !h->base->have_pq) return fail(VLQ_ERR_STATE, "PQ centroids not set");
    return VLQ_OK;
}

// nearest centroid + line + lambda for device-resident x; results in ws_near/ws_line/ws_lamf
int assign_dev(vlq_line_t h, int64_t n, const float* xd) {
    vlq_ivfpq_t b = h->base;
    TRY(h->ws_near.reserve((size_t)n * 8));
    TRY(h->ws_line.reserve((size_t)n * 4));
    TRY(h->ws_lamf.reserve((size_t)n * 4));
    TRY(b->ws_misc.reserve((size_t)n * 4));
    const int64_t page = vlq::coarse_argmin_ok(b->nlist, b->d) ? 32768 : query_page(b);   // 1-NN: no distance matrix
    for (int64_t i0 = 0; i0 < n; i0 += page) {
        const int64_t ni = std::min(page, n - i0);
        // quantizer 1-NN with true distances (classifyAndAddVectors: query(vecs, 1, ..., true))
        TRY(coarse_page(b, ni, xd + i0 * b->d, 1, b->ws_misc.as<float>() + i0,
                        h->ws_near.as<int64_t>() + i0, false, false));
    }
    vlq::launch_line_assign(xd, n, b->d, b->coarse.as<float>(), h->ws_near.as<int64_t>(),
                            h->edge_info.as<int32_t>(), h->edge_dist.as<float>(), h->nedge,
                            h->ws_line.as<int32_t>(), h->ws_lamf.as<float>(), b->stream);
    HIP_TRY(hipGetLastError());
    return VLQ_OK;
}

// full encode on device: ws_line, ws_lamb, ws_codes
int encode_dev(vlq_line_t h, int64_t n, const float* xd) {
    vlq_ivfpq_t b = h->base;
    TRY(assign_dev(h, n, xd));
    TRY(h->ws_lamb.reserve((size_t)n));
    TRY(h->ws_res.reserve((size_t)n * b->d * 4));
    TRY(h->ws_codes.reserve((size_t)n * b->M));
    vlq::launch_lambda_quantize(h->ws_lamf.as<float>(), n, h->lambda_info.as<float>(), h->nlambda,
                                h->ws_lamb.as<uint8_t>(), b->stream);
    vlq::launch_line_residuals(xd, n, b->d, b->coarse.as<float>(), h->edge_info.as<int32_t>(), h->nedge,
                               h->ws_line.as<int32_t>(), h->ws_lamb.as<uint8_t>(),
                               h->lambda_info.as<float>(), h->ws_res.as<float>(), b->stream);
    // PQ code of the residual: first minimum per sub-quantizer (ProductQuantizer.cpp:311-336)
    vlq::launch_residual_encode(h->ws_res.as<float>(), n, b->d, b->coarse.as<float>(), nullptr, 0,
                                b->pq.as<float>(), b->M, b->ksub, b->dsub, h->ws_codes.as<uint8_t>(),
                                b->stream);
    HIP_TRY(hipGetLastError());
    return VLQ_OK;
}

// la * sum(term 4) of every stored code, in the precision of the tables the next scan uses.  Returns VLQ_OK with *have =
// false when the 4 bytes per stored code cannot be had (the caller then scans with the stored term-2 rows, row mode 1, which
// needs no extra memory: same results).  Only ONE precision is kept: the other one's buffer goes back when the tables' precision
// is toggled (4 GB each at 10^9 codes).  newcnt != nullptr: only the vectors an in-place append has just added (lists.h).
int ensure_consts(vlq_line_t h, bool fp16, bool* have, const int* newcnt = nullptr) {
    vlq_ivfpq_t b = h->base;
    bool& valid = fp16 ? h->pconsth_valid : h->pconst_valid;
    *have = true;
    if (valid && !newcnt) return VLQ_OK;
    DevBuf& buf = fp16 ? h->pconsth : h->pconst;
    DevBuf& other = fp16 ? h->pconst : h->pconsth;
    if (other.p) { other.release(); (fp16 ? h->pconst_valid : h->pconsth_valid) = false; }
    if (newcnt && buf.cap < h->codes.cap / (size_t)b->M * 4 + 16) newcnt = nullptr;      // (a grown buffer starts empty: everything)
    if (buf.reserve(h->codes.cap / (size_t)b->M * 4 + 16) != VLQ_OK) {      // one float per code slot of the current layout
        (void)hipGetLastError();
        valid = false;
        *have = false;
        return VLQ_OK;
    }
    vlq::launch_line_consts(h->codes.as<uint8_t>(), h->lambdas.as<uint8_t>(), h->line_off.as<int64_t>(),
                            h->line_len.as<int64_t>(), h->edge_info.as<int32_t>(), b->term2.as<float>(),
                            fp16 ? h->term2h.as<uint16_t>() : nullptr, h->lambda_info.as<float>(), h->nedge, b->M, b->ksub,
                            h->nlines, buf.as<float>(), b->stream, newcnt);
    HIP_TRY(hipGetLastError());
    valid = true;
    return VLQ_OK;
}

void drop_consts(vlq_line_t h) { h->pconst_valid = h->pconsth_valid = false; }

}  // namespace

static vlq::ListStore line_store(vlq_line_t h) {
    vlq::ListStore ls;
    ls.nlist = h->nlines; ls.code_size = h->base->M;
    ls.codes = &h->codes; ls.lambdas = &h->lambdas; ls.ids = &h->ids; ls.off = &h->line_off; ls.len = &h->line_len;
    ls.h_off = &h->h_line_off; ls.h_len = &h->h_line_len; ls.h_stale = &h->h_lines_stale;
    return ls;
}

extern "C" {

int vlq_line_create(vlq_line_t* out, int device, int d, int nlist, int M, int nbits, int nedge,
                    int nlambda) {
    if (!out) return fail(VLQ_ERR_INVALID, "null out");
    *out = nullptr;
    if (nedge < 1 || nedge >= nlist) return fail(VLQ_ERR_INVALID, "nedge=%d must be in 1..nlist-1", nedge);
    if (nedge + 1 > VLQ_MAX_NPROBE) return fail(VLQ_ERR_UNSUPPORTED, "nedge > 1023");
    if (nlambda < 1 || nlambda > 256) return fail(VLQ_ERR_INVALID, "nlambda=%d outside 1..256 (one byte)", nlambda);
    if ((int64_t)nlist * nedge >= (int64_t(1) << 31)) return fail(VLQ_ERR_UNSUPPORTED, "nlist*nedge >= 2^31");
    vlq_line_s* h = new (std::nothrow) vlq_line_s();
    if (!h) return fail(VLQ_ERR_INVALID, "out of memory");
    int rc = vlq_ivfpq_create(&h->base, device, d, nlist, M, nbits);
    if (rc != VLQ_OK) { delete h; return rc; }
    h->nedge = nedge; h->nlambda = nlambda; h->nlines = (int64_t)nlist * nedge;
    h->h_lines_stale = true;    // host copies are filled on first use
    rc = h->line_off.reserve(((size_t)h->nlines + 1) * 8);
    if (rc == VLQ_OK) rc = h->line_len.reserve((size_t)h->nlines * 8);
    if (rc == VLQ_OK) rc = h->stats.reserve(64);
    if (rc == VLQ_OK) rc = h->codes.reserve(16);
    if (rc == VLQ_OK) rc = h->lambdas.reserve(16);
    if (rc == VLQ_OK) rc = h->ids.reserve(16);
    if (rc != VLQ_OK) { vlq_line_destroy(h); return rc; }
    (void)hipMemsetAsync(h->line_off.p, 0, ((size_t)h->nlines + 1) * 8, h->base->stream);
    (void)hipMemsetAsync(h->line_len.p, 0, (size_t)h->nlines * 8, h->base->stream);
    (void)hipMemsetAsync(h->stats.p, 0, 64, h->base->stream);
    (void)hipStreamSynchronize(h->base->stream);
    *out = h;
    return VLQ_OK;
}

void vlq_line_destroy(vlq_line_t h) {
    if (!h) return;
    if (h->base) { (void)hipSetDevice(h->base->device); (void)hipStreamSynchronize(h->base->stream); }
    DevBuf* bufs[] = {&h->edge_info, &h->edge_dist, &h->lambda_info, &h->codes, &h->lambdas, &h->ids,
                      &h->line_off, &h->line_len, &h->ws_append.cnt, &h->ws_append.cstart, &h->ws_append.keys_in,
                      &h->ws_append.keys_out, &h->ws_append.sort_tmp, &h->ws_near, &h->ws_line, &h->ws_lamf, &h->ws_lamb, &h->ws_res,
                      &h->ws_codes, &h->ws_sel_line, &h->ws_sel_b2, &h->ws_sel_g, &h->ws_sel_meta, &h->ws_sel_cnt, &h->ws_x, &h->ws_D,
                      &h->ws_I, &h->ws_keys, &h->ws_cdis, &h->stats, &h->term2h, &h->ws_qtabh, &h->pconst, &h->pconsth, &h->ws_part_keys};
    for (auto b : bufs) b->release();
    for (auto& p : h->prof_pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto e : h->prof_pool) (void)hipEventDestroy(e);
    if (h->base) vlq_ivfpq_destroy(h->base);
    delete h;
}

int vlq_line_set_stream(vlq_line_t h, void* s) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    return vlq_ivfpq_set_stream(h->base, s);
}

int vlq_line_set_row_mode(vlq_line_t h, int mode) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (mode < 0 || mode > 3) return fail(VLQ_ERR_INVALID, "row mode %d outside 0..3", mode);
    h->row_mode = mode;
    return VLQ_OK;
}

int vlq_line_set_scan_parts(vlq_line_t h, int parts) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (parts < 0 || parts > 64) return fail(VLQ_ERR_INVALID, "scan parts %d outside 0..64", parts);
    h->scan_parts = parts;
    return VLQ_OK;
}

int vlq_line_set_float16_tables(vlq_line_t h, int enable) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (enable && !(h->base->M == 16 && h->base->ksub == 256))
        return fail(VLQ_ERR_UNSUPPORTED, "float16 look-up tables are built for 16 x 8-bit codes (the reference drivers' shape)");
    h->fp16_tables = enable != 0;
    return VLQ_OK;
}

int vlq_line_set_coarse_centroids(vlq_line_t h, const float* c) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    h->have_graph = false;
    h->term2h_valid = false;
    drop_consts(h);
    return vlq_ivfpq_set_coarse_centroids(h->base, c);
}

int vlq_line_set_pq_centroids(vlq_line_t h, const float* c) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    h->term2h_valid = false;
    drop_consts(h);
    return vlq_ivfpq_set_pq_centroids(h->base, c);
}

int vlq_line_set_lambda_codebook(vlq_line_t h, const float* li) {
    if (!h || !li) return fail(VLQ_ERR_INVALID, "null argument");
    TRY(set_dev(h->base));
    drop_consts(h);
    TRY(h->lambda_info.reserve(256 * 4));            // the 16-byte scan copies all 256 slots into LDS
    HIP_TRY(hipMemsetAsync(h->lambda_info.p, 0, 256 * 4, h->base->stream));
    HIP_TRY(hipMemcpyAsync(h->lambda_info.p, li, (size_t)h->nlambda * 4, hipMemcpyDefault, h->base->stream));
    h->h_lambda.resize(h->nlambda);
    HIP_TRY(hipMemcpyAsync(h->h_lambda.data(), h->lambda_info.p, (size_t)h->nlambda * 4, hipMemcpyDeviceToHost, h->base->stream));
    HIP_TRY(hipStreamSynchronize(h->base->stream));
    h->have_lambda = true;
    return VLQ_OK;
}

int vlq_line_set_graph(vlq_line_t h, const int32_t* ei, const float* ed) {
    if (!h || !ei || !ed) return fail(VLQ_ERR_INVALID, "null argument");
    TRY(set_dev(h->base));
    drop_consts(h);
    const size_t n = (size_t)h->nlines;
    std::vector<int32_t> chk(n);
    HIP_TRY(hipMemcpy(chk.data(), ei, n * 4, hipMemcpyDefault));
    for (size_t i = 0; i < n; i++)
        if (chk[i] < 0 || chk[i] >= h->base->nlist) return fail(VLQ_ERR_INVALID, "edge_info[%zu]=%d out of range", i, chk[i]);
    TRY(h->edge_info.reserve(n * 4));
    TRY(h->edge_dist.reserve(n * 4));
    HIP_TRY(hipMemcpyAsync(h->edge_info.p, chk.data(), n * 4, hipMemcpyHostToDevice, h->base->stream));
    HIP_TRY(hipMemcpyAsync(h->edge_dist.p, ed, n * 4, hipMemcpyDefault, h->base->stream));
    HIP_TRY(hipStreamSynchronize(h->base->stream));
    h->have_graph = true;
    return VLQ_OK;
}

int vlq_line_build_graph(vlq_line_t h, int32_t* ei_out, float* ed_out) {
    TRY(need(h, false, false, false));
    vlq_ivfpq_t b = h->base;
    TRY(set_dev(b));
    const int nl = b->nlist, E = h->nedge, k = E + 1;
    // self-query of the centroid set for nedge+1 neighbours (buildGraphNonPaged_, :869-893)
    std::vector<float> D((size_t)nl * k);
    std::vector<int64_t> I((size_t)nl * k);
    TRY(h->ws_cdis.reserve((size_t)nl * k * 4));
    TRY(h->ws_keys.reserve((size_t)nl * k * 8));
    const int64_t page = query_page(b);
    for (int64_t i0 = 0; i0 < nl; i0 += page) {
        const int64_t ni = std::min<int64_t>(page, nl - i0);
        TRY(coarse_page(b, ni, b->coarse.as<float>() + i0 * b->d, k, h->ws_cdis.as<float>() + i0 * k,
                        h->ws_keys.as<int64_t>() + i0 * k, false, false));
    }
    HIP_TRY(hipMemcpyAsync(D.data(), h->ws_cdis.p, D.size() * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(I.data(), h->ws_keys.p, I.size() * 8, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    std::vector<int32_t> ei((size_t)nl * E);
    std::vector<float> ed((size_t)nl * E);
    for (int i = 0; i < nl; i++)
        for (int e = 0; e < E; e++) {   // first column (the centroid itself) dropped
            ei[(size_t)i * E + e] = (int32_t)I[(size_t)i * k + e + 1];
            ed[(size_t)i * E + e] = D[(size_t)i * k + e + 1];
        }
    TRY(vlq_line_set_graph(h, ei.data(), ed.data()));
    if (ei_out) memcpy(ei_out, ei.data(), ei.size() * 4);
    if (ed_out) memcpy(ed_out, ed.data(), ed.size() * 4);
    return VLQ_OK;
}

int vlq_line_assign(vlq_line_t h, int64_t n, const float* x, int32_t* line_id, float* lambdaf) {
    TRY(need(h, true, false, false));
    if (n < 0 || (n > 0 && (!x || !line_id || !lambdaf))) return fail(VLQ_ERR_INVALID, "bad argument");
    if (n == 0) return VLQ_OK;
    vlq_ivfpq_t b = h->base;
    TRY(set_dev(b));
    const void* xd;
    TRY(stage_in(b, x, (size_t)n * b->d * 4, h->ws_x, &xd));
    TRY(assign_dev(h, n, (const float*)xd));
    HIP_TRY(hipMemcpyAsync(line_id, h->ws_line.p, (size_t)n * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(lambdaf, h->ws_lamf.p, (size_t)n * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return VLQ_OK;
}

int vlq_line_residuals(vlq_line_t h, int64_t n, const float* x, float* residuals) {
    TRY(need(h, true, true, false));
    if (n < 0 || (n > 0 && (!x || !residuals))) return fail(VLQ_ERR_INVALID, "bad argument");
    if (n == 0) return VLQ_OK;
    vlq_ivfpq_t b = h->base;
    TRY(set_dev(b));
    const void* xd;
    TRY(stage_in(b, x, (size_t)n * b->d * 4, h->ws_x, &xd));
    TRY(assign_dev(h, n, (const float*)xd));
    TRY(h->ws_lamb.reserve((size_t)n));
    TRY(h->ws_res.reserve((size_t)n * b->d * 4));
    vlq::launch_lambda_quantize(h->ws_lamf.as<float>(), n, h->lambda_info.as<float>(), h->nlambda,
                                h->ws_lamb.as<uint8_t>(), b->stream);
    vlq::launch_line_residuals((const float*)xd, n, b->d, b->coarse.as<float>(), h->edge_info.as<int32_t>(),
                               h->nedge, h->ws_line.as<int32_t>(), h->ws_lamb.as<uint8_t>(),
                               h->lambda_info.as<float>(), h->ws_res.as<float>(), b->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(residuals, h->ws_res.p, (size_t)n * b->d * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return VLQ_OK;
}

int vlq_line_encode(vlq_line_t h, int64_t n, const float* x, int32_t* line_id, uint8_t* lambda,
                    uint8_t* codes) {
    TRY(need(h, true, true, true));
    if (n < 0 || (n > 0 && (!x || !line_id || !lambda || !codes))) return fail(VLQ_ERR_INVALID, "bad argument");
    if (n == 0) return VLQ_OK;
    vlq_ivfpq_t b = h->base;
    TRY(set_dev(b));
    const void* xd;
    TRY(stage_in(b, x, (size_t)n * b->d * 4, h->ws_x, &xd));
    TRY(encode_dev(h, n, (const float*)xd));
    HIP_TRY(hipMemcpyAsync(line_id, h->ws_line.p, (size_t)n * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(lambda, h->ws_lamb.p, (size_t)n, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(codes, h->ws_codes.p, (size_t)n * b->M, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return VLQ_OK;
}

int vlq_line_set_lists(vlq_line_t h, const uint8_t* codes, const uint8_t* lambdas, const int64_t* ids,
                       const int64_t* line_offsets) {
    if (!h || !line_offsets) return fail(VLQ_ERR_INVALID, "null argument");
    vlq_ivfpq_t b = h->base;
    TRY(set_dev(b));
    drop_consts(h);
    std::vector<int64_t> off((size_t)h->nlines + 1);
    HIP_TRY(hipMemcpy(off.data(), line_offsets, off.size() * 8, hipMemcpyDefault));
    if (off[0] != 0) return fail(VLQ_ERR_INVALID, "line_offsets[0] != 0");
    for (int64_t i = 0; i < h->nlines; i++)
        if (off[i + 1] < off[i]) return fail(VLQ_ERR_INVALID, "line_offsets not monotone at %ld", (long)i);
    const int64_t nt = off[h->nlines];
    if (nt > 0 && (!codes || !lambdas || !ids)) return fail(VLQ_ERR_INVALID, "null codes/lambdas/ids");
    TRY(h->codes.reserve((size_t)nt * b->M + 16));
    TRY(h->lambdas.reserve((size_t)nt + 16));
    TRY(h->ids.reserve((size_t)nt * 8 + 16));
    if (nt > 0) {
        HIP_TRY(hipMemcpyAsync(h->codes.p, codes, (size_t)nt * b->M, hipMemcpyDefault, b->stream));
        HIP_TRY(hipMemcpyAsync(h->lambdas.p, lambdas, (size_t)nt, hipMemcpyDefault, b->stream));
        HIP_TRY(hipMemcpyAsync(h->ids.p, ids, (size_t)nt * 8, hipMemcpyDefault, b->stream));
    }
    std::vector<int64_t> len((size_t)h->nlines);
    for (int64_t i = 0; i < h->nlines; i++) len[(size_t)i] = off[(size_t)i + 1] - off[(size_t)i];
    HIP_TRY(hipMemcpyAsync(h->line_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(h->line_len.p, len.data(), len.size() * 8, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    h->h_line_off.swap(off);
    h->h_line_len.swap(len);
    h->h_lines_stale = false;
    h->ntotal = nt;
    h->ntotal_added = nt;        // sequential ids continue from the loaded count (IndexIVFPQ.cpp:244)
    return VLQ_OK;
}

int vlq_line_add(vlq_line_t h, int64_t n, const float* x, const int64_t* xids) {
    TRY(need(h, true, true, true));
    if (n < 0 || (n > 0 && !x)) return fail(VLQ_ERR_INVALID, "bad argument");
    if (n == 0) return VLQ_OK;
    vlq_ivfpq_t b = h->base;
    TRY(set_dev(b));
    // assign + encode + append on the device; the reference keeps the encoded batch on the host
    // until it is written out (gpu/GpuIndexIVFPQ.cu:852-857)
    const void* xd;
    TRY(stage_in(b, x, (size_t)n * b->d * 4, h->ws_x, &xd));
    TRY(encode_dev(h, n, (const float*)xd));
    const void* idd = nullptr;
    if (xids) TRY(stage_in(b, xids, (size_t)n * 8, h->ws_keys, &idd));
    vlq::ListStore ls = line_store(h);
    int64_t placed = 0;
    bool relaid = false;
    const int rc = vlq::lists_append(ls, h->ws_append, n, nullptr, h->ws_line.as<int32_t>(), h->ws_codes.as<uint8_t>(),
                                     h->ws_lamb.as<uint8_t>(), (const int64_t*)idd, h->ntotal_added, b->stream, &placed, &relaid);
    if (rc != VLQ_OK) { drop_consts(h); return rc; }
    h->ntotal += placed;         // vectors without a line are dropped
    h->ntotal_added += n;
    // the stored codes' constants (line16c.hip): an append in place adds the new vectors' only -- O(batch), not O(database), per
    // add; a rebuilt layout (amortised: 25 % slack) moves every list, the constants are then rebuilt before the next search
    if (relaid || !b->term2_valid) drop_consts(h);
    else {
        for (int fp16 = 0; fp16 < 2; fp16++) {
            if (!(fp16 ? h->pconsth_valid : h->pconst_valid)) continue;
            bool have = false;
            TRY(ensure_consts(h, fp16 != 0, &have, h->ws_append.cnt.as<int>()));
        }
    }
    return VLQ_OK;
}

int64_t vlq_line_ntotal(vlq_line_t h) { return h ? h->ntotal : -1; }

int vlq_line_list_length(vlq_line_t h, int64_t line, int64_t* len) {
    if (!h || !len) return fail(VLQ_ERR_INVALID, "null argument");
    if (line < 0 || line >= h->nlines) return fail(VLQ_ERR_INVALID, "line id out of range");
    if (h->h_lines_stale) {
        TRY(set_dev(h->base));
        vlq::ListStore ls = line_store(h);
        TRY(vlq::lists_sync_host(ls, h->base->stream));
    }
    *len = h->h_line_len[line];
    return VLQ_OK;
}

int vlq_line_get_list(vlq_line_t h, int64_t line, uint8_t* codes_out, uint8_t* lambdas_out, int64_t* ids_out) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    if (line < 0 || line >= h->nlines) return fail(VLQ_ERR_INVALID, "line id out of range");
    vlq_ivfpq_t b = h->base;
    TRY(set_dev(b));
    {
        vlq::ListStore ls = line_store(h);
        TRY(vlq::lists_sync_host(ls, b->stream));
    }
    const int64_t o = h->h_line_off[line], len = h->h_line_len[line];
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (len > 0 && codes_out) HIP_TRY(hipMemcpy(codes_out, h->codes.as<uint8_t>() + o * b->M, (size_t)len * b->M, hipMemcpyDeviceToHost));
    if (len > 0 && lambdas_out) HIP_TRY(hipMemcpy(lambdas_out, h->lambdas.as<uint8_t>() + o, (size_t)len, hipMemcpyDeviceToHost));
    if (len > 0 && ids_out) HIP_TRY(hipMemcpy(ids_out, h->ids.as<int64_t>() + o, (size_t)len * 8, hipMemcpyDeviceToHost));
    return VLQ_OK;
}

int vlq_line_search(vlq_line_t h, int64_t n, const float* x, int nprobe, int w1, int k, float* D,
                    int64_t* I, int32_t* lines_out) {
    TRY(need(h, true, true, true));
    if (n < 0 || (n > 0 && (!x || !D || !I))) return fail(VLQ_ERR_INVALID, "bad argument");
    if (nprobe < 1 || nprobe > VLQ_MAX_NPROBE) return fail(VLQ_ERR_INVALID, "nprobe=%d outside 1..1024", nprobe);
    if (w1 < 1 || w1 > 1024) return fail(VLQ_ERR_INVALID, "w1=%d outside 1..1024", w1);
    if (k < 1 || k > VLQ_MAX_K) return fail(VLQ_ERR_INVALID, "k=%d outside 1..1024", k);
    if (n == 0) return VLQ_OK;
    vlq_ivfpq_t b = h->base;
    TRY(set_dev(b));
    TRY(vlq_ivfpq_set_search_options(b, 1, 1, 0));
    nprobe = std::min(nprobe, b->nlist);          // IVFPQ.cu:702
    // rows rebuilt in the scan kernel (line16r.hip) need no term-2 table at all
    const bool rebuilt_rows = h->row_mode == 2 && !h->fp16_tables && b->M == 16 && b->ksub == 256 && k <= 256 &&
                              (b->dsub == 4 || b->dsub == 6 || b->dsub == 8) && (int64_t)nprobe * h->nedge < (int64_t(1) << 24);
    if (!rebuilt_rows) TRY(ensure_term2(b));
    const size_t E = (size_t)b->M * b->ksub;
    const void* xd;
    TRY(stage_in(b, x, (size_t)n * b->d * 4, h->ws_x, &xd));
    void *Dd, *Id;
    bool copyD, copyI;
    TRY(stage_out(D, (size_t)n * k * 4, h->ws_D, &Dd, &copyD));
    TRY(stage_out(I, (size_t)n * k * 8, h->ws_I, &Id, &copyI));
    const int64_t page = query_page(b);
    const int64_t pn = std::min(n, page);
    TRY(h->ws_keys.reserve((size_t)pn * nprobe * 8));
    TRY(h->ws_cdis.reserve((size_t)pn * nprobe * 4));
    TRY(h->ws_sel_line.reserve((size_t)n * w1 * 4));
    TRY(h->ws_sel_b2.reserve((size_t)pn * w1 * 4));
    TRY(h->ws_sel_g.reserve((size_t)pn * w1 * 4));
    TRY(h->ws_sel_meta.reserve((size_t)pn * w1 * sizeof(vlq::LineMeta)));
    TRY(h->ws_sel_cnt.reserve((size_t)pn * 4));
    TRY(b->ws_qtab.reserve((size_t)pn * E * 4));
    for (int64_t i0 = 0; i0 < n; i0 += page) {
        const int64_t ni = std::min(page, n - i0);
        const float* xi = (const float*)xd + i0 * b->d;
        // compact records (16-byte scan kernel): their sort key holds the candidate index in 24 bits
        const bool small_tables = (b->M & 3) == 0 && b->M <= 32 && b->M * b->ksub <= 2048;
        const bool with_meta = ((b->M == 16 && b->ksub == 256) || small_tables) && (int64_t)nprobe * h->nedge < (int64_t(1) << 24);
        const bool fp16 = h->fp16_tables && with_meta && b->M == 16 && b->ksub == 256;
        if (fp16 && !h->term2h_valid) {      // half(term 2), once per trained state (impl/IVFPQ.cu:1442 toHalf)
            TRY(h->term2h.reserve((size_t)b->nlist * E * 2));
            vlq::launch_to_half(b->term2.as<float>(), (int64_t)b->nlist * (int64_t)E, 1.f, h->term2h.as<uint16_t>(), b->stream);
            h->term2h_valid = true;
        }
        int32_t* sel_line = h->ws_sel_line.as<int32_t>() + i0 * w1;
        // (Steps 1 and 2 in sub-pages whose distance matrix stays in the 256 MB Infinity Cache between the kernel that writes
        // it and the line select that gathers from it were measured and lose: 2000 queries in pages of 500 / 1000 queries 3.22 /
        // 3.06 ms against 2.98 for one page -- the smaller launches cost more than the cache returns.)
        const int64_t sub = ni;
        static const bool ls_old = getenv("VLQ_LINE_SELECT_WAVE") != nullptr;     // A/B: the one-wave-per-query kernel
        for (int64_t j0 = 0; j0 < ni; j0 += sub) {
            const int64_t nj = std::min(sub, ni - j0);
            // 1. all centroid "distances" without |q|^2 + the nprobe nearest (Distance.cu:233-383)
            TRY(coarse_page(b, nj, xi + j0 * b->d, nprobe, h->ws_cdis.as<float>(), h->ws_keys.as<int64_t>(), true, false, true));
            // 2. the w1 best lines among nprobe x nedge (BroadcastSum.cu:477-560)
            vlq::LineMeta* metaj = with_meta ? h->ws_sel_meta.as<vlq::LineMeta>() + j0 * w1 : nullptr;
            if (!ls_old && vlq::line_select2_supports(nprobe, h->nedge, w1))
                vlq::launch_line_select2(b->ws_dist.as<float>(), nj, b->nlist, h->ws_keys.as<int64_t>(), nprobe,
                                         h->edge_info.as<int32_t>(), h->edge_dist.as<float>(), h->nedge, w1,
                                         sel_line + j0 * w1, h->ws_sel_b2.as<float>() + j0 * w1, h->ws_sel_g.as<float>() + j0 * w1,
                                         b->stream, h->line_off.as<int64_t>(), h->line_len.as<int64_t>(), VLQ_LINE_MAX_CODES,
                                         metaj, h->ws_sel_cnt.as<int32_t>() + j0);
            else
                vlq::launch_line_select(b->ws_dist.as<float>(), nj, b->nlist, h->ws_keys.as<int64_t>(), nprobe,
                                        h->edge_info.as<int32_t>(), h->edge_dist.as<float>(), h->nedge, w1,
                                        sel_line + j0 * w1, h->ws_sel_b2.as<float>() + j0 * w1, h->ws_sel_g.as<float>() + j0 * w1,
                                        b->stream, h->line_off.as<int64_t>(), h->line_len.as<int64_t>(), VLQ_LINE_MAX_CODES,
                                        metaj, h->ws_sel_cnt.as<int32_t>() + j0);
        }
        // the stored codes' share of the distance (line16c.hip), once per database state
        bool use_consts = !rebuilt_rows && (h->row_mode == 0 || h->row_mode == 3) && with_meta && b->M == 16 &&
                          b->ksub == 256 && w1 <= 1024 && h->ntotal > 0;
        if (use_consts) TRY(ensure_consts(h, fp16, &use_consts));        // (no memory for them: the stored-row kernel below)
        // 3. per-query <q_m, cent_mj> (term 3 / -2, IVFPQ.cu:1409-1432)
        vlq::launch_pq_tables(xi, ni, b->d, b->pq.as<float>(), b->M, b->ksub, b->dsub, nullptr, 0,
                              b->ws_qtab.as<float>(), b->stream);
        if (fp16) {                            // half(term 3) = half(-2 <q_m, cent_mj>)
            TRY(h->ws_qtabh.reserve((size_t)pn * E * 2));
            vlq::launch_to_half(b->ws_qtab.as<float>(), ni * (int64_t)E, -2.f, h->ws_qtabh.as<uint16_t>(), b->stream);
        }
        // 4. scan + top-k
        vlq::LineScanArgs a;
        a.codes = h->codes.as<uint8_t>(); a.lambdas = h->lambdas.as<uint8_t>(); a.ids = h->ids.as<int64_t>();
        a.line_off = h->line_off.as<int64_t>(); a.line_len = h->line_len.as<int64_t>(); a.term2 = rebuilt_rows ? nullptr : b->term2.as<float>(); a.qtab = b->ws_qtab.as<float>();
        a.edge_info = h->edge_info.as<int32_t>(); a.edge_dist = h->edge_dist.as<float>();
        a.lambda_info = h->lambda_info.as<float>();
        a.sel_line = sel_line; a.sel_b2 = h->ws_sel_b2.as<float>(); a.sel_g = h->ws_sel_g.as<float>();
        if (fp16) { a.term2h = h->term2h.as<uint16_t>(); a.qtabh = h->ws_qtabh.as<uint16_t>(); }
        a.sel_meta = with_meta ? h->ws_sel_meta.as<vlq::LineMeta>() : nullptr; a.sel_cnt = h->ws_sel_cnt.as<int32_t>();
        a.D = (float*)Dd + i0 * k; a.I = (int64_t*)Id + i0 * k;
        a.ncode = h->stats.as<unsigned long long>();
        a.nq = ni; a.w1 = w1; a.k = k; a.M = b->M; a.ksub = b->ksub; a.nedge = h->nedge;
        a.max_line_codes = VLQ_LINE_MAX_CODES;
        a.nprobe = nprobe;
        if (use_consts) {
            a.pconst = fp16 ? h->pconsth.as<float>() : h->pconst.as<float>();
            a.nparts = h->scan_parts > 0 ? h->scan_parts : vlq::line16c_parts(ni, k, 8);
            if (a.nparts > 1) {
                TRY(h->ws_part_keys.reserve((size_t)ni * a.nparts * k * 8));
                a.part_keys = h->ws_part_keys.as<unsigned long long>();
            }
        }
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (h->prof) {
            auto get = [&]() { hipEvent_t e = nullptr; if (!h->prof_pool.empty()) { e = h->prof_pool.back(); h->prof_pool.pop_back(); } else if (hipEventCreate(&e) != hipSuccess) e = nullptr; return e; };
            ev0 = get(); ev1 = get();
            if (ev0 && ev1) (void)hipEventRecord(ev0, b->stream);
        }
        if (rebuilt_rows) {
            a.coarse = b->coarse.as<float>(); a.pq_cent = b->pq.as<float>(); a.pq_rnorm = b->rnorm.as<float>();
            a.term2 = nullptr;
            vlq::launch_line16r_scan(a, b->dsub, b->stream);
        } else if (use_consts && vlq::line16c_supports(a)) {
            vlq::launch_line16c_scan(a, b->stream);
        } else {
            vlq::launch_line_scan(a, b->stream);
        }
        if (ev0 && ev1) { (void)hipEventRecord(ev1, b->stream); h->prof_pending.push_back({ev0, ev1}); }
        HIP_TRY(hipGetLastError());
    }
    if (lines_out) HIP_TRY(hipMemcpyAsync(lines_out, h->ws_sel_line.p, (size_t)n * w1 * 4, hipMemcpyDeviceToHost, b->stream));
    TRY(finish_outputs(b, copyD, D, Dd, (size_t)n * k * 4, copyI, I, Id, (size_t)n * k * 8));
    if (lines_out) HIP_TRY(hipStreamSynchronize(b->stream));
    return VLQ_OK;
}

int vlq_line_profile(vlq_line_t h, int enable) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    h->prof = enable != 0;
    return VLQ_OK;
}

int vlq_line_profile_read(vlq_line_t h, double* scan_ms, int64_t* launches, int reset) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    TRY(set_dev(h->base));
    for (auto& p : h->prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            h->prof_ms += ms;
            h->prof_calls++;
        }
        h->prof_pool.push_back(p.first);
        h->prof_pool.push_back(p.second);
    }
    h->prof_pending.clear();
    if (scan_ms) *scan_ms = h->prof_ms;
    if (launches) *launches = h->prof_calls;
    if (reset) { h->prof_ms = 0; h->prof_calls = 0; }
    return VLQ_OK;
}

int vlq_line_stats(vlq_line_t h, uint64_t* ncode, int reset) {
    if (!h) return fail(VLQ_ERR_INVALID, "null handle");
    vlq_ivfpq_t b = h->base;
    TRY(set_dev(b));
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(st, h->stats.p, 64, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (ncode) *ncode = st[0];
    if (getenv("VLQ_L16C_TIMING") && st[5])
        fprintf(stderr, "[l16c timing] per workgroup: prologue %.2f us, loop %.2f us, tail %.2f us (%llu workgroups)\n",
                st[2] * 0.01 / st[5], st[3] * 0.01 / st[5], st[4] * 0.01 / st[5], st[5]);
    if (reset) HIP_TRY(hipMemsetAsync(h->stats.p, 0, 64, b->stream));
    return VLQ_OK;
}

}  // extern "C"
