// Host-side launch interface of the VLQ kernels (line.hip).
#pragma once
#include "kernels.h"

namespace vlq {

void launch_line_assign(const float* x, int64_t n, int d, const float* coarse, const int64_t* nearest,
                        const int32_t* edge_info, const float* edge_dist, int nedge, int32_t* line_id,
                        float* lambdaf, hipStream_t s);
void launch_lambda_quantize(const float* lambdaf, int64_t n, const float* lambda_info, int nlambda,
                            uint8_t* out, hipStream_t s);
void launch_line_residuals(const float* x, int64_t n, int d, const float* coarse, const int32_t* edge_info,
                           int nedge, const int32_t* line_id, const uint8_t* lambda,
                           const float* lambda_info, float* res, hipStream_t s);
// one selected, non-empty line as the 16-byte scan kernel reads it (48 bytes = three 16-byte loads)
struct LineMeta {
    int64_t off;      // first code of the line
    int32_t len;      // codes scanned (capped at max_line_codes)
    int32_t line;     // line id = c * nedge + e
    int32_t s;        // far-end centroid
    float c2;         // |s - c|^2
    float b2;         // coarse value of the anchor c
    float g;          // v[s] - v[c]
    uint32_t pos0;    // scan position of the line's first code: lines in the order the line select emits
                      // them (ascending key, BroadcastSum.cu:538-553), codes in list order
    int32_t rank;     // index of the line among the query's non-empty kept lines in that order
    int32_t anchor;   // c = line / nedge
    int32_t pad1;
};
static_assert(sizeof(LineMeta) == 48, "LineMeta is read as three 16-byte words");

// sel_meta / sel_cnt (optional): compact LineMeta records [nq][w1] + their count per query
void launch_line_select(const float* dist, int64_t nq, int nlist, const int64_t* keys, int nprobe,
                        const int32_t* edge_info, const float* edge_dist, int nedge, int w1,
                        int32_t* sel_line, float* sel_b2, float* sel_g, hipStream_t s,
                        const int64_t* line_off = nullptr, const int64_t* line_len = nullptr,
                        int max_line_codes = 0, LineMeta* sel_meta = nullptr, int32_t* sel_cnt = nullptr);

// the same selection and outputs by one workgroup per query (line_select2.hip): radix threshold + one sort of the winners
bool line_select2_supports(int nprobe, int nedge, int w1);
void launch_line_select2(const float* dist, int64_t nq, int nlist, const int64_t* keys, int nprobe, const int32_t* edge_info,
                         const float* edge_dist, int nedge, int w1, int32_t* sel_line, float* sel_b2, float* sel_g, hipStream_t s,
                         const int64_t* line_off, const int64_t* line_len, int max_line_codes, LineMeta* sel_meta, int32_t* sel_cnt);

struct LineScanArgs {
    const uint8_t* codes;        // [ntotal][M] line-contiguous
    const uint8_t* lambdas;      // [ntotal]
    const int64_t* ids;          // [ntotal]
    const int64_t* line_off;     // [nlist*nedge + 1] line starts
    const int64_t* line_len = nullptr;   // [nlist*nedge] lengths, or nullptr: packed (lists.h)
    const float* term2;          // [nlist][M*ksub]
    const float* qtab;           // [nq][M*ksub]  <q_m, cent_mj>
    // float16 look-up tables (useFloat16LookupTables, 16-byte scan only): half(term2) and half(-2 <q_m, cent_mj>)
    const uint16_t* term2h = nullptr;    // [nlist][M*ksub]
    const uint16_t* qtabh = nullptr;     // [nq][M*ksub]
    // row-recomputing 16-byte scan (line16r.hip): coarse centroids, PQ codebook [M][ksub][dsub] and its norms
    const float* coarse = nullptr;       // [nlist][d]
    const float* pq_cent = nullptr;      // [M][ksub][dsub]
    const float* pq_rnorm = nullptr;     // [M][ksub]
    const int32_t* edge_info;    // [nlist*nedge]
    const float* edge_dist;      // [nlist*nedge]
    const float* lambda_info;    // [nlambda]
    const int32_t* sel_line;     // [nq][w1]
    const float* sel_b2;         // [nq][w1]
    const float* sel_g;          // [nq][w1]
    // per-code constants la * sum_m term4[m][code_m] (line16c.hip), in the precision of the tables in use
    const float* pconst = nullptr;        // [capacity]
    int nprobe = 0;                       // anchors per query (sizes the scan's group tables; 0: w1)
    unsigned long long* part_keys = nullptr;   // [nq][nparts][k] (distance, scan position) keys of a query's parts
    int nparts = 1;                       // workgroups per query (line16c_parts)
    const LineMeta* sel_meta = nullptr;   // [nq][w1] compact records (16-byte scan)
    const int32_t* sel_cnt = nullptr;     // [nq]
    float* D;
    int64_t* I;
    unsigned long long* ncode;
    int64_t nq;
    int w1, k, M, ksub, nedge, max_line_codes;
};
void launch_line_scan(const LineScanArgs& a, hipStream_t s);
// 16-byte codes, term-2 rows rebuilt in registers instead of read (line16r.hip): same results as
// launch_line_scan, bit for bit
bool line16r_supports(const LineScanArgs& a, int dsub);
void launch_line16r_scan(const LineScanArgs& a, int dsub, hipStream_t s);
// per-code constants of the stored codes (line16c.hip): out[i] = la_i * sum_m (term2[s][m][c_m] - term2[c][m][c_m]) for
// every code of every line, with the scan kernels' operations; term2h != nullptr: the float16 tables' form
void launch_line_consts(const uint8_t* codes, const uint8_t* lambdas, const int64_t* line_off, const int64_t* line_len,
                        const int32_t* edge_info, const float* term2, const uint16_t* term2h, const float* lambda_info,
                        int nedge, int M, int ksub, int64_t nlines, float* out, hipStream_t s,
                        const int* newcnt = nullptr);
// 16-byte codes with the stored constants: one table per anchor centroid, no far-end rows; same results as
// launch_line_scan, bit for bit
bool line16c_supports(const LineScanArgs& a);
int line16c_parts(int64_t nq, int k, int max_parts);
void launch_line16c_scan(const LineScanArgs& a, hipStream_t s);
// out[i] = half(scale * in[i]) (round to nearest even); scale = 1 (term 2) or -2 (term 3, IVFPQ.cu:1409-1442)
void launch_to_half(const float* in, int64_t n, float scale, uint16_t* out, hipStream_t s);

}  // namespace vlq
