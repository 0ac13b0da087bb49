/*
 * vlq_line.h -- C ABI of the fork's "vector and line quantization" (VLQ) index on
 * MI355X: the path behind the reference's
 *   GpuIndexIVFPQ(resources, dims, nlist, subQuantizers, bitsPerCode, nedge, nLambda,
 *                 metric, config)                      gpu/GpuIndexIVFPQ.h:60-68
 * i.e. train / buildGraph_ / classifyAndAddVectors / searchImpl_ -> IVFPQ::queryGraph
 * (gpu/GpuIndexIVFPQ.cu:346-403, :577-908, :1400-1460; gpu/impl/IVFPQ.cu:685-775).
 *
 * Data model: coarse centroids c_i with a k-NN graph (`nedge` edges per centroid); a
 * vector lives on line id = i*nedge + e (from c_i towards s = edge_info[i][e]) with a
 * one-byte index into the scalar codebook lambda_info[nlambda] and an M-byte PQ code
 * of its residual to the anchor (1-l) c_i + l s.  Lines are the inverted lists.
 *
 * Same conventions as vlq_ivfpq.h ([h|d] pointers, status codes, vlq_last_error()).
 * Returned distances omit |q|^2 exactly as the fork's search does
 * (gpu/impl/Distance.cu:286-291): D = |q - y|^2 - |q|^2.
 */
#ifndef VLQ_LINE_H
#define VLQ_LINE_H

#include "vlq_ivfpq.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vlq_line_s* vlq_line_t;

#define VLQ_LINE_MAX_CODES 1024   /* codes visited per line, PQScanMultiPassPrecomputed.cu:728 */

int vlq_line_create(vlq_line_t* out, int device, int d, int nlist, int M, int nbits, int nedge,
                    int nlambda);
void vlq_line_destroy(vlq_line_t h);
int vlq_line_set_stream(vlq_line_t h, void* hip_stream);

/* trained state (GpuIndexIVFPQ::readCodebookFromFile layout: centroids | pq | edgeInfo |
 * edgeDistInfo | lambdaInfo, gpu/GpuIndexIVFPQ.cu:1731-1758).  All [h|d]. */
int vlq_line_set_coarse_centroids(vlq_line_t h, const float* centroids);      /* [nlist*d] */
int vlq_line_set_pq_centroids(vlq_line_t h, const float* centroids);          /* [M][ksub][dsub] */
int vlq_line_set_lambda_codebook(vlq_line_t h, const float* lambda_info);     /* [nlambda] */
int vlq_line_set_graph(vlq_line_t h, const int32_t* edge_info, const float* edge_dist); /* [nlist*nedge] */
/* GpuIndexFlat::buildGraph (gpu/GpuIndexFlat.cu:375-429,869-893): nedge nearest other
 * centroids of every centroid, on the device.  Outputs optional host buffers. */
int vlq_line_build_graph(vlq_line_t h, int32_t* edge_info_out, float* edge_dist_out);

/* training primitives (trainResidualQuantizer_, gpu/GpuIndexIVFPQ.cu:346-403):
 *   assign   : nearest centroid + best line + real-valued lambda (get1BinKernel_nms)
 *   residuals: x - ((1-l) c + l s) with l = lambda_info[quantised lambda] (calResidual)
 * x [h|d]; outputs host buffers. */
int vlq_line_assign(vlq_line_t h, int64_t n, const float* x, int32_t* line_id, float* lambdaf);
int vlq_line_residuals(vlq_line_t h, int64_t n, const float* x, float* residuals);
/* full encoding: line id, lambda byte, PQ code (classifyAndAddVectors :577-908) */
int vlq_line_encode(vlq_line_t h, int64_t n, const float* x, int32_t* line_id, uint8_t* lambda,
                    uint8_t* codes);

/* add_with_ids (addImpl_ -> classifyAndAddVectors); xids may be NULL */
int vlq_line_add(vlq_line_t h, int64_t n, const float* x, const int64_t* xids);
/* bulk load, line-contiguous (readDbFromFile: .dbcodes / .dblas / .dbIdx / .dbcount,
 * gpu/GpuIndexIVFPQ.cu:1813-1844): codes[ntotal][M], lambdas[ntotal], ids[ntotal],
 * line_offsets[nlist*nedge+1].  [h|d] */
int vlq_line_set_lists(vlq_line_t h, const uint8_t* codes, const uint8_t* lambdas,
                       const int64_t* ids, const int64_t* line_offsets);
int64_t vlq_line_ntotal(vlq_line_t h);
int vlq_line_list_length(vlq_line_t h, int64_t line, int64_t* len);
int vlq_line_get_list(vlq_line_t h, int64_t line, uint8_t* codes_out, uint8_t* lambdas_out,
                      int64_t* ids_out);

/* GpuIndexIVFPQ::search with nprobe_ coarse centroids and w1_ lines kept per query
 * (queryGraph).  nprobe <= 1024, w1 <= 1024, k <= 1024.  x, D, I [h|d].
 * lines_out (optional, host, [n*w1]): the selected line ids in selection order. */
int vlq_line_search(vlq_line_t h, int64_t n, const float* x, int nprobe, int w1, int k, float* D,
                    int64_t* I, int32_t* lines_out);
int vlq_line_stats(vlq_line_t h, uint64_t* ncode, int reset);
/* Scan-kernel time of the searches since the last reset, measured with HIP events recorded on the index's stream
 * around every scan launch (what bench.py prices against the roofline; the role of the reference's KernelTimer,
 * gpu/utils/Timer.h:19-54).  enable: 0 off, 1 on.  *scan_ms = sum over launches, *launches = their number. */
int vlq_line_profile(vlq_line_t h, int enable);
int vlq_line_profile_read(vlq_line_t h, double* scan_ms, int64_t* launches, int reset);

/* GpuIndexIVFPQConfig::useFloat16LookupTables (gpu/GpuIndexIVFPQ.h:24-38) for the VLQ search -- what the
 * reference's VLQ drivers run with (gpu/test/deep1b16_query.cpp:239-243).  As in the reference: term 2 and
 * term 3 are kept as half (impl/IVFPQ.cu:1442), the per-line tables are formed in half arithmetic
 * (impl/PQScanMultiPassPrecomputed.cu:54-75, :313-334) and the looked-up entries are summed in float.
 * Off by default (fp32 tables); M = 16 x 8 bit only.  Halves the table bytes that bound the scan.
 * As in the reference, the table entries must fit the half range (|value| <= 65504): fine for
 * normalised descriptors (Deep1B, the drivers' data), NOT for raw byte-valued vectors such as SIFT,
 * whose term 2 entries reach 10^5 and become infinite. */
int vlq_line_set_float16_tables(vlq_line_t h, int enable);

/* Where the 16-byte scan takes a line's term-2 rows from (speed / memory only: the results are bit-identical):
 *   0, 3 = the query-independent half of a code's distance, la * sum_m term4[m][code_m] (impl/PQScanMultiPassPrecomputed.cu:
 *       783-811; term4 = term2[s] - term2[c] depends on the line a code is stored on, not on the query), is computed once
 *       per database state with the scan's own operations and read back as 4 bytes per stored code (line16c.hip; M = 16 x 8
 *       bit): no far-end row, one table per anchor centroid, half the look-ups.  0 = "automatic" resolves to this.
 *   1 = rows read from the stored table per line, as the reference's kernel reads them
 *   2 = rows rebuilt in registers from the far-end centroid and the PQ codebook (line16r.hip; M = 16 x 8 bit,
 *       dsub in {4, 6, 8}, k <= 256, fp32 tables -- other shapes keep the stored rows): no term-2 table in HBM
 *       at all (1 GiB less at 65 536 centroids) and a quarter of the scan's HBM traffic (11 against 45 GB per
 *       2000 queries at the reference driver's geometry), paid for with 2.7x the VALU work: 8.8 against 7.1 ms
 *       there.  For memory-constrained deployments; not the default. */
int vlq_line_set_row_mode(vlq_line_t h, int mode);
/* Workgroups per query of the scan with stored per-code constants (row mode 0 / 3): a query's kept codes are cut
 * into `parts` equal ranges scanned by separate workgroups whose candidates a merge kernel joins under the total
 * order (distance, scan position) -- results do not depend on it.  0 = automatic: the count that fills the chip's
 * last round best (small batches, e.g. the per-GPU slice of a sharded batch, get several).  1..64. */
int vlq_line_set_scan_parts(vlq_line_t h, int parts);

#ifdef __cplusplus
}
#endif
#endif /* VLQ_LINE_H */
