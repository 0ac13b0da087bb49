// =============================================================================
// ORACLE (VLQ row) -- TEST INFRASTRUCTURE ONLY.  Same rules as ivfpq_oracle.cpp:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// CPU restatement of the fork's "vector and line quantization" path (SURVEY.md §8a
// row a11).  The reference implements it ONLY as CUDA kernels; there is no CPU
// version and nothing in this container can run CUDA, so:
//
//     PARITY UNPINNED against the reference binary.
//
// The restatement is derived from the kernels' arithmetic (citations below) and is
// pinned only by self-consistency tests (tests/test_vlq_oracle.py): returned
// distances equal  |q - ((1-l)c + l s) - r|^2 - |q|^2  recomputed in float64 from
// the decoded vectors, the chosen line minimises the point-to-segment distance, etc.
// Where the CUDA code leaves the order of operations or tie-breaking to the compiler
// or to an unstable sort (bitonic3, BlockSelect), THIS FILE fixes one (unfused
// multiplies/adds left to right, ties to the lowest index) and the HIP path is held to
// it bit for bit.
//
// Notation (gpu/GpuIndexIVFPQ.h:58-81): coarse centroids c_i, a k-NN graph with
// `nedge` edges per centroid: edge (i, e) goes to s = edge_info[i][e] with squared
// length edge_dist[i][e]; a database vector is stored on line id = i*nedge + e with a
// one-byte index into the 1-D codebook lambda_info[nlambda] and a PQ code of the
// residual to the anchor (1-l) c_i + l s.
// =============================================================================
#include <immintrin.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <vector>

extern "C" {

float orc_fvec_L2sqr(const float* x, const float* y, size_t d);
float orc_fvec_inner_product(const float* x, const float* y, size_t d);
float orc_fvec_norm_L2sqr(const float* x, size_t d);
void orc_knn_L2sqr(const float* x, const float* y, size_t d, size_t nx, size_t ny, size_t k,
                   float* D, int64_t* I, int canonical, int force_path);

struct orc_vlq {
    int32_t d, nlist, M, nbits, ksub, dsub, nedge, nlambda;
    const float* coarse;        // [nlist][d]
    const float* pq_centroids;  // [M][ksub][dsub]
    const int32_t* edge_info;   // [nlist][nedge]
    const float* edge_dist;     // [nlist][nedge]
    const float* lambda_info;   // [nlambda]
    const float* term2;         // [nlist][M][ksub]  (= IndexIVFPQ precomputed table, impl/IVFPQ.cu:599-684)
    const uint8_t* codes;       // [ntotal][M]   line-contiguous
    const uint8_t* lambdas;     // [ntotal]
    const int64_t* ids;         // [ntotal]
    const int64_t* line_off;    // [nlist*nedge + 1]
};

// geometry helpers, gpu/utils/triangle.cuh:54-87 (float, unfused, left to right)
static inline float vlq_project(float a2, float b2, float c2) { return -0.5f * (a2 - b2 - c2) / c2; }
static inline float vlq_dist2(float a2, float b2, float c2, float l) {
    return (b2 + (l * l) * c2) + l * (a2 - b2 - c2);
}

// GpuIndexFlat::buildGraph (gpu/GpuIndexFlat.cu:375-429, :869-893): k+1 nearest centroids of
// every centroid, first column (the centroid itself) dropped.
void orc_vlq_build_graph(const float* coarse, int nlist, int d, int nedge, int32_t* edge_info,
                         float* edge_dist) {
    std::vector<float> D((size_t)nlist * (nedge + 1));
    std::vector<int64_t> I((size_t)nlist * (nedge + 1));
    orc_knn_L2sqr(coarse, coarse, d, nlist, nlist, nedge + 1, D.data(), I.data(), 1, 2);
    for (int i = 0; i < nlist; i++)
        for (int e = 0; e < nedge; e++) {
            edge_info[(size_t)i * nedge + e] = (int32_t)I[(size_t)i * (nedge + 1) + e + 1];
            edge_dist[(size_t)i * nedge + e] = D[(size_t)i * (nedge + 1) + e + 1];
        }
}

// get1BinKernel_nms (gpu/GpuIndexFlat.cu:433-557): for the nearest centroid A of x and each
// of its edges: a2 = |x - s|^2, b2 = |x - c_A|^2, c2 = edge length; lambda = project, d2 =
// dist2; pick the edge of smallest d2 among those with 0 <= lambda <= 1, else the smallest
// overall.  Ties -> lowest edge (the reference's bitonic sort leaves them unspecified).
void orc_vlq_assign(const orc_vlq* ix, const float* x, size_t n, const int64_t* nearest,
                    int32_t* line_id, float* lambdaf) {
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        const int64_t A = nearest[i];
        if (A < 0) { line_id[i] = -1; lambdaf[i] = 0.f; continue; }
        const float* xi = x + i * ix->d;
        const float b2 = orc_fvec_L2sqr(xi, ix->coarse + (size_t)A * ix->d, ix->d);
        int best = -1, best_in = -1;
        float bd = 0.f, bd_in = 0.f, bl = 0.f, bl_in = 0.f;
        for (int e = 0; e < ix->nedge; e++) {
            const int s = ix->edge_info[(size_t)A * ix->nedge + e];
            const float a2 = orc_fvec_L2sqr(xi, ix->coarse + (size_t)s * ix->d, ix->d);
            const float c2 = ix->edge_dist[(size_t)A * ix->nedge + e];
            const float l = vlq_project(a2, b2, c2);
            const float d2 = vlq_dist2(a2, b2, c2, l);
            if (best < 0 || d2 < bd) { best = e; bd = d2; bl = l; }
            if (l >= 0.f && l <= 1.f && (best_in < 0 || d2 < bd_in)) { best_in = e; bd_in = d2; bl_in = l; }
        }
        const int e = best_in >= 0 ? best_in : best;
        line_id[i] = (int32_t)(A * ix->nedge + e);
        lambdaf[i] = best_in >= 0 ? bl_in : bl;
    }
}

// assignLambdaKernel (gpu/GpuIndexFlat.cu:559-602): nearest scalar of the codebook, first minimum
void orc_vlq_quantize_lambda(const orc_vlq* ix, const float* lambdaf, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; i++) {
        int bi = 0;
        float bd = FLT_MAX;
        for (int j = 0; j < ix->nlambda; j++) {
            const float t = lambdaf[i] - ix->lambda_info[j];
            const float dd = t * t;
            if (dd < bd) { bd = dd; bi = j; }
        }
        out[i] = (uint8_t)bi;
    }
}

// calResidual (gpu/GpuIndexFlat.cu:1092-1129): x - ((1-l) c_A + l c_s), l = lambda_info[byte]
void orc_vlq_residuals(const orc_vlq* ix, const float* x, size_t n, const int32_t* line_id,
                       const uint8_t* lambda, float* res) {
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        if (line_id[i] < 0) { memset(res + i * ix->d, 0, sizeof(float) * ix->d); continue; }
        const int A = line_id[i] / ix->nedge;
        const int s = ix->edge_info[line_id[i]];
        const float la = ix->lambda_info[lambda[i]];
        const float oml = 1.f - la;
        const float* ca = ix->coarse + (size_t)A * ix->d;
        const float* cs = ix->coarse + (size_t)s * ix->d;
        for (int j = 0; j < ix->d; j++)
            res[i * ix->d + j] = x[i * ix->d + j] - (oml * ca[j] + la * cs[j]);
    }
}

// PQ encode of the residuals: ProductQuantizer::compute_code semantics (first minimum)
void orc_vlq_pq_encode(const orc_vlq* ix, const float* res, size_t n, uint8_t* codes) {
#pragma omp parallel for
    for (size_t i = 0; i < n; i++)
        for (int m = 0; m < ix->M; m++) {
            float mind = 1e20f;
            int bi = -1;
            for (int j = 0; j < ix->ksub; j++) {
                const float dd = orc_fvec_L2sqr(res + i * ix->d + m * ix->dsub,
                                                ix->pq_centroids + ((size_t)m * ix->ksub + j) * ix->dsub, ix->dsub);
                if (dd < mind) { mind = dd; bi = j; }
            }
            codes[i * ix->M + m] = (uint8_t)bi;
        }
}

// --- float16 look-up tables (config.useFloat16LookupTables, what the reference's VLQ drivers run with:
// gpu/test/deep1b16_query.cpp:239-243).  The reference keeps term 2 and term 3 as half
// (impl/IVFPQ.cu:1442 toHalf; precompTerm2 / precompTerm3 are Tensor<half>), forms the two LDS tables
// with HALF arithmetic (loadPrecomputedTerm :54-75 Math<Half8>::add, loadPrecomputedTermGraph :313-334
// Math<Half8>::sub) and accumulates the looked-up entries in float (ConvertTo<float>::to, :798-805).
// h16(): float -> half, round to nearest even.  hadd / hsub: IEEE half add = the float sum of the two
// (exactly representable operands) rounded once more to half -- innocuous double rounding, because
// float carries 24 >= 2*11 + 2 significand bits.
static inline uint16_t h16(float x) { return (uint16_t)_cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT); }
static inline float f32h(uint16_t h) { return _cvtsh_ss(h); }
static inline uint16_t hadd16(uint16_t a, uint16_t b) { return h16(f32h(a) + f32h(b)); }
static inline uint16_t hsub16(uint16_t a, uint16_t b) { return h16(f32h(a) - f32h(b)); }

struct VCand { float dis; int64_t pos; int64_t id; };
static inline bool vless(const VCand& a, const VCand& b) { return a.dis < b.dis || (a.dis == b.dis && a.pos < b.pos); }

// Search (IVFPQ::queryGraph impl/IVFPQ.cu:685-775):
//  1. coarse: v_j = |c_j|^2 - 2 <q, c_j> for all centroids (the fork omits |q|^2,
//     impl/Distance.cu:286-291), inner product = k-ordered fmaf chain as in the IVFPQ
//     oracle; the nprobe smallest (v, j).
//  2. line select (sumAlongRowsWithOrder2, impl/BroadcastSum.cu:477-560): candidate i =
//     rank*nedge + e over the nprobe centroids x their edges; a2 = v[s], b2 = v[c],
//     c2 = edge length, g = a2 - b2, t = g - c2, key = t > 0 ? b2 : b2 - 0.25*t*t/c2;
//     the w1 smallest (key, i), emitted in that (ascending) order (:538-553).
//  3. scan (pqScanPrecomputedMultiPassGraph, impl/PQScanMultiPassPrecomputed.cu:675-811):
//     per selected line (c, s), at most 1024 codes (:728); per code with l = lambda_info[byte],
//     in the order the source writes it (:783-811):
//        dist = (b2 + l*g) + (l*l - l)*c2
//        dist += T23[m][code_m], m = 0..M-1   with T23 = term2[c] + (-2)<q_m, cent>  (:54-75)
//        tmp  += T4[m][code_m]  (from 0)      with T4  = term2[s] - term2[c]          (:313-334)
//        out = dist + l*tmp
//  4. the k smallest (dist, scan position) with the heap's strict-< admission against
//     FLT_MAX; labels are the stored ids.  Scan position = offset in the reference's output
//     array: lines in their emitted (ascending key) order, codes in list order.
//
// ORDER CHOICES this file makes where the CUDA source leaves them open (every one is also what the
// HIP kernels do, bit for bit):
//   a. every a*b+c is an unfused multiply then add (nvcc's default -fmad=true would contract some of
//      them; which ones is a compiler decision, so the written operator order is followed instead);
//   b. the coarse inner product is a k-ascending fmaf chain (the reference calls cuBLAS);
//      <q_m, cent> of term3 likewise is the IVFPQ oracle's SSE-order inner product (cuBLAS there);
//   c. |x - c|^2 of the assignment is the SSE-order sum (the reference tree-reduces across a block);
//   d. equal keys: lowest candidate index in the line select, lowest edge in the assignment, first
//      minimum in the lambda quantiser, lowest scan position in the final top-k (the reference's
//      bitonic3 / BlockSelect leave ties unspecified).
// Returns the number of codes visited.
static int64_t vlq_search_impl(const orc_vlq* ix, const float* xq, size_t nq, int nprobe, int w1, int k,
                               float* D, int64_t* I, int32_t* lines_out, int fp16);

int64_t orc_vlq_search(const orc_vlq* ix, const float* xq, size_t nq, int nprobe, int w1, int k,
                       float* D, int64_t* I, int32_t* lines_out /* [nq][w1] or NULL */) {
    return vlq_search_impl(ix, xq, nq, nprobe, w1, k, D, I, lines_out, 0);
}
// the same search with float16 look-up tables (see h16 above); coarse stage and line select are fp32
int64_t orc_vlq_search_fp16(const orc_vlq* ix, const float* xq, size_t nq, int nprobe, int w1, int k,
                            float* D, int64_t* I, int32_t* lines_out) {
    return vlq_search_impl(ix, xq, nq, nprobe, w1, k, D, I, lines_out, 1);
}

static int64_t vlq_search_impl(const orc_vlq* ix, const float* xq, size_t nq, int nprobe, int w1, int k,
                               float* D, int64_t* I, int32_t* lines_out, int fp16) {
    const int d = ix->d, E = ix->nedge;
    const size_t mk = (size_t)ix->M * ix->ksub;
    std::vector<float> cn(ix->nlist);
    for (int j = 0; j < ix->nlist; j++) cn[j] = orc_fvec_norm_L2sqr(ix->coarse + (size_t)j * d, d);
    std::vector<float> cT((size_t)d * ix->nlist);
    for (int j = 0; j < ix->nlist; j++)
        for (int c = 0; c < d; c++) cT[(size_t)c * ix->nlist + j] = ix->coarse[(size_t)j * d + c];
    int64_t ncode = 0;
#pragma omp parallel reduction(+ : ncode)
    {
        std::vector<float> v(ix->nlist), t3(mk);
        std::vector<VCand> cand, lines, res;
#pragma omp for
        for (size_t qi = 0; qi < nq; qi++) {
            const float* q = xq + qi * d;
            for (int j = 0; j < ix->nlist; j++) v[j] = 0.f;
            for (int c = 0; c < d; c++) {
                const float xc = q[c];
                const float* row = &cT[(size_t)c * ix->nlist];
                for (int j = 0; j < ix->nlist; j++) v[j] = fmaf(xc, row[j], v[j]);
            }
            for (int j = 0; j < ix->nlist; j++) v[j] = (0.f + cn[j]) - 2 * v[j];
            cand.clear();
            for (int j = 0; j < ix->nlist; j++) if (v[j] < FLT_MAX) cand.push_back({v[j], j, j});
            const int np = std::min<int>(nprobe, (int)cand.size());
            std::partial_sort(cand.begin(), cand.begin() + np, cand.end(), vless);
            lines.clear();
            for (int r = 0; r < np; r++) {
                const int c = (int)cand[r].id;
                const float b2 = v[c];
                for (int e = 0; e < E; e++) {
                    const int s = ix->edge_info[(size_t)c * E + e];
                    const float c2 = ix->edge_dist[(size_t)c * E + e];
                    const float g = v[s] - b2;
                    const float t = g - c2;
                    const float key = (t > 0) ? b2 : b2 - 0.25f * t * t / c2;
                    if (key < FLT_MAX) lines.push_back({key, (int64_t)r * E + e, (int64_t)c * E + e});
                }
            }
            const int nw = std::min<int>(w1, (int)lines.size());
            // the kept lines are emitted -- and therefore laid out in the scan's output array
            // (prefixSumOffsets, PQScanMultiPassPrecomputed.cu:711-712) -- in ascending key order
            // (BroadcastSum.cu:538-553: the sorted heap contents); scan positions follow this order
            std::partial_sort(lines.begin(), lines.begin() + nw, lines.end(), vless);
            for (int m = 0; m < ix->M; m++)
                for (int j = 0; j < ix->ksub; j++)
                    t3[m * ix->ksub + j] = -2.0f * orc_fvec_inner_product(
                        q + m * ix->dsub, ix->pq_centroids + ((size_t)m * ix->ksub + j) * ix->dsub, ix->dsub);
            res.clear();
            int64_t pos = 0;
            for (int w = 0; w < nw; w++) {
                const int64_t line = lines[w].id;
                if (lines_out) lines_out[qi * w1 + w] = (int32_t)line;
                const int c = (int)(line / E), e = (int)(line % E);
                const int s = ix->edge_info[(size_t)c * E + e];
                const float b2 = v[c], g = v[s] - v[c], c2 = ix->edge_dist[(size_t)c * E + e];
                const float* t2c = ix->term2 + (size_t)c * mk;
                const float* t2s = ix->term2 + (size_t)s * mk;
                const int64_t o = ix->line_off[line];
                const int64_t len = std::min<int64_t>(ix->line_off[line + 1] - o, 1024);
                for (int64_t jj = 0; jj < len; jj++) {
                    const uint8_t* code = ix->codes + (size_t)(o + jj) * ix->M;
                    const float l = ix->lambda_info[ix->lambdas[o + jj]];
                    // PQScanMultiPassPrecomputed.cu:783-811, in the order written:
                    //   dist = term1 + la*term6 + (la*la-la)*term5;  dist += term23[m] (m ascending);
                    //   tmp += term4[m];  out = dist + la*tmp
                    float dist = (b2 + l * g) + (l * l - l) * c2;
                    float tmp = 0.f;
                    for (int m = 0; m < ix->M; m++) {
                        const size_t idx = (size_t)m * ix->ksub + code[m];
                        if (fp16) {
                            const uint16_t hc = h16(t2c[idx]), hs = h16(t2s[idx]);
                            dist += f32h(hadd16(hc, h16(t3[idx])));
                            tmp += f32h(hsub16(hs, hc));
                            continue;
                        }
                        dist += t2c[idx] + t3[idx];          // term23 entry (loadPrecomputedTerm :54-75)
                        tmp += t2s[idx] - t2c[idx];          // term4 entry (loadPrecomputedTermGraph :313-334)
                    }
                    dist = dist + l * tmp;
                    if (dist < FLT_MAX) res.push_back({dist, pos + jj, ix->ids[o + jj]});
                }
                pos += len;
                ncode += len;
            }
            if (lines_out) for (int w = nw; w < w1; w++) lines_out[qi * w1 + w] = -1;
            const int nk = std::min<int>(k, (int)res.size());
            std::partial_sort(res.begin(), res.begin() + nk, res.end(), vless);
            for (int i = 0; i < nk; i++) { D[qi * k + i] = res[i].dis; I[qi * k + i] = res[i].id; }
            for (int i = nk; i < k; i++) { D[qi * k + i] = FLT_MAX; I[qi * k + i] = -1; }
        }
    }
    return ncode;
}

}  // extern "C"
