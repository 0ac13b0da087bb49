// =============================================================================
// ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's IVFPQ search/add hot path (SURVEY.md §8a
// rows a1-a10).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library, and only as the checker / the timed
// CPU baseline -- never as (part of) the product path.
//
// Parity status: PINNED.  Every function below is checked bit-for-bit against
// outputs of the reference's own CPU library (built by oracle/ref.mk from
// /root/reference, driven by oracle/ref_driver.cpp) through the committed
// fixtures tests/golden/*.npz (tests/test_oracle_golden.py), with one
// documented exception: the coarse-quantizer inner products for batches of
// >= 20 queries.  The reference computes those with BLAS sgemm_
// (utils.cpp:869), whose summation order is vendor-defined and unpinned by the
// reference (makefile.inc lists MKL/OpenBLAS/ATLAS).  Here they are a k-ordered
// fmaf chain (== the gfx950 f32 MFMA accumulation order), so on that stage the
// oracle matches the reference to rounding (tested: same probe sets, coarse_dis
// within 1e-5 relative) and everything downstream is compared at the
// search_knn_with_key seam with shared (keys, coarse_dis), where it is
// bit-exact.
//
// Written from the reference's algorithmic description; file:line citations are
// relative to /root/reference.  Scalar code reproduces the SSE lane order of the
// reference's intrinsics; compile with -ffp-contract=off.
// =============================================================================
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <immintrin.h>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

// Plain-data view of an IndexIVFPQ (IndexIVFPQ.h:29-47, IndexIVF.h:45-59,
// ProductQuantizer.h:25-60) with the inverted lists stored list-contiguously.
struct orc_index {
    int32_t d, nlist, M, nbits;
    int32_t ksub, dsub, code_size;
    int32_t by_residual;            // IndexIVFPQ.h:30
    int32_t use_precomputed_table;  // IndexIVFPQ.h:31 (0 or 1; 2 = IMI not restated)
    int32_t float16_tables;         // != 0: GpuIndexIVFPQConfig::useFloat16LookupTables for the table-1 scan (below)
    int64_t max_codes;              // IndexIVFPQ.h:40 (0 = unlimited)
    const float* coarse_centroids;  // [nlist][d]   IndexFlat::xb
    const float* pq_centroids;      // [M][ksub][dsub] ProductQuantizer.h:51-60
    const float* precomputed_table; // [nlist][M][ksub] or NULL
    const uint8_t* codes;           // [ntotal][code_size], list-major
    const int64_t* ids;             // [ntotal], list-major
    const int64_t* list_offsets;    // [nlist+1]
    // inverted multi-index coarse quantizer (MultiIndexQuantizer, IndexPQ.h:124-160):
    // imi_nbits > 0 -> nlist = 2^(imi_M*imi_nbits), key = sum_m idx_m << (m*imi_nbits)
    int32_t imi_M, imi_nbits;
    const float* imi_centroids;     // [imi_M][2^imi_nbits][d/imi_M]
};

// ---------------------------------------------------------------------------
// SSE-order scalar kernels (utils.cpp:481-556).  Four lane accumulators, one
// non-fused multiply and add per element, tail through a zero-padded read,
// then hadd(hadd()) = (s0+s1)+(s2+s3).
// ---------------------------------------------------------------------------
float orc_fvec_inner_product(const float* x, const float* y, size_t d) {
    // utils.cpp:509-533 (tail product is added unconditionally, :523-528)
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    while (d >= 4) {
        for (int l = 0; l < 4; l++) s[l] = s[l] + x[l] * y[l];
        x += 4; y += 4; d -= 4;
    }
    for (int l = 0; l < 4; l++) {
        float mx = (size_t)l < d ? x[l] : 0.f, my = (size_t)l < d ? y[l] : 0.f;
        s[l] = s[l] + mx * my;
    }
    return (s[0] + s[1]) + (s[2] + s[3]);
}

float orc_fvec_norm_L2sqr(const float* x, size_t d) {
    // utils.cpp:538-556
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    while (d >= 4) {
        for (int l = 0; l < 4; l++) s[l] = s[l] + x[l] * x[l];
        x += 4; d -= 4;
    }
    for (int l = 0; l < 4; l++) {
        float mx = (size_t)l < d ? x[l] : 0.f;
        s[l] = s[l] + mx * mx;
    }
    return (s[0] + s[1]) + (s[2] + s[3]);
}

float orc_fvec_L2sqr(const float* x, const float* y, size_t d) {
    // utils.cpp:481-506 (tail only when d > 0, :496-502)
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    while (d >= 4) {
        for (int l = 0; l < 4; l++) { float a = x[l] - y[l]; s[l] = s[l] + a * a; }
        x += 4; y += 4; d -= 4;
    }
    if (d > 0) {
        for (int l = 0; l < 4; l++) {
            float mx = (size_t)l < d ? x[l] : 0.f, my = (size_t)l < d ? y[l] : 0.f;
            float a = mx - my;
            s[l] = s[l] + a * a;
        }
    }
    return (s[0] + s[1]) + (s[2] + s[3]);
}

void orc_fvec_norms_L2sqr(float* nr, const float* x, size_t d, size_t nx) {
    // utils.cpp:675-682
#pragma omp parallel for
    for (size_t i = 0; i < nx; i++) nr[i] = orc_fvec_norm_L2sqr(x + i * d, d);
}

// c = a + bf * b, multiply and add not fused (utils.cpp:1832-1863)
static inline void fvec_madd(size_t n, const float* a, float bf, const float* b, float* c) {
    for (size_t i = 0; i < n; i++) c[i] = a[i] + bf * b[i];
}

// ---------------------------------------------------------------------------
// Binary max-heap on (val, id) arrays: Heap.h:89-143 (pop/push), :186-209
// (heapify with no initial elements), :296-323 (reorder).  1-based sifting.
// ---------------------------------------------------------------------------
static inline void maxheap_pop(size_t k, float* bh_val, int64_t* bh_ids) {
    bh_val--; bh_ids--;
    float val = bh_val[k];
    size_t i = 1, i1, i2;
    while (1) {
        i1 = i << 1; i2 = i1 + 1;
        if (i1 > k) break;
        if (i2 == k + 1 || bh_val[i1] > bh_val[i2]) {
            if (val > bh_val[i1]) break;
            bh_val[i] = bh_val[i1]; bh_ids[i] = bh_ids[i1]; i = i1;
        } else {
            if (val > bh_val[i2]) break;
            bh_val[i] = bh_val[i2]; bh_ids[i] = bh_ids[i2]; i = i2;
        }
    }
    bh_val[i] = bh_val[k]; bh_ids[i] = bh_ids[k];
}

static inline void maxheap_push(size_t k, float* bh_val, int64_t* bh_ids, float val, int64_t id) {
    bh_val--; bh_ids--;
    size_t i = k, i_father;
    while (i > 1) {
        i_father = i >> 1;
        if (!(val > bh_val[i_father])) break;
        bh_val[i] = bh_val[i_father]; bh_ids[i] = bh_ids[i_father]; i = i_father;
    }
    bh_val[i] = val; bh_ids[i] = id;
}

static inline void maxheap_heapify(size_t k, float* bh_val, int64_t* bh_ids) {
    for (size_t i = 0; i < k; i++) { bh_val[i] = FLT_MAX; bh_ids[i] = -1; }
}

static inline size_t maxheap_reorder(size_t k, float* bh_val, int64_t* bh_ids) {
    size_t i, ii;
    for (i = 0, ii = 0; i < k; i++) {
        float val = bh_val[0];
        int64_t id = bh_ids[0];
        maxheap_pop(k - i, bh_val, bh_ids);
        bh_val[k - ii - 1] = val;
        bh_ids[k - ii - 1] = id;
        if (id != -1) ii++;
    }
    size_t nel = ii;
    memmove(bh_val, bh_val + k - ii, ii * sizeof(*bh_val));
    memmove(bh_ids, bh_ids + k - ii, ii * sizeof(*bh_ids));
    for (; ii < k; ii++) { bh_val[ii] = FLT_MAX; bh_ids[ii] = -1; }
    return nel;
}

// Canonical selection used to check the GPU path: the k smallest under the total
// order (distance, scan position), which equals the heap's result up to
// permutations inside groups of exactly equal distance (SURVEY.md §7 "tie
// breaking"; same rule as the reference's own select test,
// gpu/test/TestGpuSelect.cu:82-114).
struct Cand { float dis; int64_t pos; int64_t id; };
static inline bool cand_less(const Cand& a, const Cand& b) {
    return a.dis < b.dis || (a.dis == b.dis && a.pos < b.pos);
}
static void canonical_topk(std::vector<Cand>& c, size_t k, float* D, int64_t* I) {
    size_t n = std::min(k, c.size());
    std::partial_sort(c.begin(), c.begin() + n, c.end(), cand_less);
    for (size_t i = 0; i < n; i++) { D[i] = c[i].dis; I[i] = c[i].id; }
    for (size_t i = n; i < k; i++) { D[i] = FLT_MAX; I[i] = -1; }
}

// ---------------------------------------------------------------------------
// Coarse quantizer: IndexFlat::search (IndexFlat.cpp:42-56) -> knn_L2sqr
// (utils.cpp:935-946).
//   nx < 20 && d%4==0 : knn_L2sqr_sse (utils.cpp:757-786), fvec_L2sqr per pair
//   otherwise         : knn_L2sqr_blas (utils.cpp:834-901):
//                       dis = (|x|^2 + |y|^2) - 2*ip, ip from sgemm_.
// `ip` here = fmaf chain over k = 0..d-1 starting from 0 (see header).
// canonical != 0 selects by (dis, j) instead of replaying the heap.
// ---------------------------------------------------------------------------
void orc_knn_L2sqr(const float* x, const float* y, size_t d, size_t nx, size_t ny,
                   size_t k, float* D, int64_t* I, int canonical, int force_path) {
    // force_path: 0 = reference dispatch, 1 = sse path, 2 = blas path
    bool sse = (d % 4 == 0 && nx < 20);
    if (force_path == 1) sse = true;
    if (force_path == 2) sse = false;
    std::vector<float> xn, yn, yT;
    if (!sse) {
        xn.resize(nx); yn.resize(ny);
        orc_fvec_norms_L2sqr(xn.data(), x, d, nx);
        orc_fvec_norms_L2sqr(yn.data(), y, d, ny);
        yT.resize(d * ny);  // [d][ny] so the fmaf chains vectorise across j
        for (size_t j = 0; j < ny; j++)
            for (size_t c = 0; c < d; c++) yT[c * ny + j] = y[j * d + c];
    }
#pragma omp parallel
    {
        std::vector<float> disv(ny);
        std::vector<Cand> cands;
#pragma omp for
        for (size_t i = 0; i < nx; i++) {
            const float* xi = x + i * d;
            if (sse) {
                for (size_t j = 0; j < ny; j++) disv[j] = orc_fvec_L2sqr(xi, y + j * d, d);
            } else {
                for (size_t j = 0; j < ny; j++) disv[j] = 0.f;
                for (size_t c = 0; c < d; c++) {
                    const float xc = xi[c];
                    const float* yr = &yT[c * ny];
                    for (size_t j = 0; j < ny; j++) disv[j] = fmaf(xc, yr[j], disv[j]);
                }
                for (size_t j = 0; j < ny; j++) disv[j] = (xn[i] + yn[j]) - 2 * disv[j];
            }
            float* simi = D + i * k;
            int64_t* idxi = I + i * k;
            if (canonical) {
                cands.clear();
                for (size_t j = 0; j < ny; j++)
                    if (disv[j] < FLT_MAX) cands.push_back({disv[j], (int64_t)j, (int64_t)j});
                canonical_topk(cands, k, simi, idxi);
            } else {
                maxheap_heapify(k, simi, idxi);
                for (size_t j = 0; j < ny; j++) {
                    if (disv[j] < simi[0]) {
                        maxheap_pop(k, simi, idxi);
                        maxheap_push(k, simi, idxi, disv[j], j);
                    }
                }
                maxheap_reorder(k, simi, idxi);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// MultiIndexQuantizer::search (IndexPQ.cpp:804-857) with MinSumK / SemiSortedArray
// (IndexPQ.cpp:524-778).  Distance tables = ProductQuantizer::compute_distance_tables
// (ProductQuantizer.cpp:438-461): sub-vectors of < 16 dims -> fvec_L2sqr per entry;
// otherwise pairwise_L2sqr (utils.cpp:1311-1355) = (|x|^2 + |y|^2) + (-2)*sgemm, whose
// inner product is restated as the k-ordered fmaf chain (BLAS order is vendor-defined,
// as for the flat coarse quantizer).  SemiSortedArray sorts table entries with an
// indirect heap whose order among EXACTLY equal values is an implementation detail;
// here equal values are ordered by index (unpinned only for exact ties).
// ---------------------------------------------------------------------------
namespace {
inline void minheap_push(size_t k, float* bh_val, int64_t* bh_ids, float val, int64_t id) {
    bh_val--; bh_ids--;
    size_t i = k, i_father;
    while (i > 1) {
        i_father = i >> 1;
        if (!(val < bh_val[i_father])) break;
        bh_val[i] = bh_val[i_father]; bh_ids[i] = bh_ids[i_father]; i = i_father;
    }
    bh_val[i] = val; bh_ids[i] = id;
}
inline void minheap_pop(size_t k, float* bh_val, int64_t* bh_ids) {
    bh_val--; bh_ids--;
    float val = bh_val[k];
    size_t i = 1, i1, i2;
    while (1) {
        i1 = i << 1; i2 = i1 + 1;
        if (i1 > k) break;
        if (i2 == k + 1 || bh_val[i1] < bh_val[i2]) {
            if (val < bh_val[i1]) break;
            bh_val[i] = bh_val[i1]; bh_ids[i] = bh_ids[i1]; i = i1;
        } else {
            if (val < bh_val[i2]) break;
            bh_val[i] = bh_val[i2]; bh_ids[i] = bh_ids[i2]; i = i2;
        }
    }
    bh_val[i] = bh_val[k]; bh_ids[i] = bh_ids[k];
}
}  // namespace

void orc_imi_distance_tables(const orc_index* ix, const float* x, size_t n, float* tabs) {
    const int Mc = ix->imi_M, kc = 1 << ix->imi_nbits, dc = ix->d / Mc;
    if (dc < 16) {
#pragma omp parallel for
        for (size_t i = 0; i < n; i++)
            for (int m = 0; m < Mc; m++)
                for (int j = 0; j < kc; j++)
                    tabs[(i * Mc + m) * kc + j] = orc_fvec_L2sqr(
                        x + i * ix->d + m * dc, ix->imi_centroids + ((size_t)m * kc + j) * dc, dc);
    } else {
        for (int m = 0; m < Mc; m++) {
            std::vector<float> bn(kc);
            for (int j = 0; j < kc; j++) bn[j] = orc_fvec_norm_L2sqr(ix->imi_centroids + ((size_t)m * kc + j) * dc, dc);
#pragma omp parallel for
            for (size_t i = 0; i < n; i++) {
                const float* xi = x + i * ix->d + m * dc;
                const float qn = orc_fvec_norm_L2sqr(xi, dc);
                for (int j = 0; j < kc; j++) {
                    const float* y = ix->imi_centroids + ((size_t)m * kc + j) * dc;
                    float ip = 0.f;
                    for (int c = 0; c < dc; c++) ip = fmaf(xi[c], y[c], ip);
                    tabs[(i * Mc + m) * kc + j] = (qn + bn[j]) + (-2.0f) * ip;
                }
            }
        }
    }
}

void orc_imi_search(const orc_index* ix, const float* x, size_t n, size_t k, float* D, int64_t* I) {
    const int Mc = ix->imi_M, kc = 1 << ix->imi_nbits, nb = ix->imi_nbits;
    std::vector<float> tabs(n * Mc * kc);
    orc_imi_distance_tables(ix, x, n, tabs.data());
#pragma omp parallel
    {
        std::vector<std::vector<int> > perm(Mc, std::vector<int>(kc));
        std::vector<float> hv(k * Mc + 1);
        std::vector<int64_t> hi(k * Mc + 1), weights(Mc);
#pragma omp for
        for (size_t q = 0; q < n; q++) {
            const float* t = &tabs[q * Mc * kc];
            if (k == 1) {   // IndexPQ.cpp:815-840
                float dis = 0;
                int64_t label = 0;
                for (int s = 0; s < Mc; s++) {
                    float vmin = HUGE_VALF;
                    int64_t lmin = -1;
                    for (int j = 0; j < kc; j++)
                        if (t[s * kc + j] < vmin) { vmin = t[s * kc + j]; lmin = j; }
                    dis += vmin;
                    label |= lmin << (s * nb);
                }
                D[q] = dis;
                I[q] = label;
                continue;
            }
            // MinSumK<float, SemiSortedArray<float>, false>::run (IndexPQ.cpp:690-778)
            weights[0] = 1;
            for (int m = 1; m < Mc; m++) weights[m] = weights[m - 1] * kc;
            for (int m = 0; m < Mc; m++) {
                for (int j = 0; j < kc; j++) perm[m][j] = j;
                const float* xm = t + m * kc;
                std::stable_sort(perm[m].begin(), perm[m].end(), [xm](int a, int b) { return xm[a] < xm[b]; });
            }
            auto val = [&](int m, int r) { return t[m * kc + perm[m][r]]; };
            float* sums = D + q * k;
            int64_t* terms = I + q * k;
            size_t heap_size = 0;
            float sum = 0;
            terms[0] = 0;
            for (int m = 0; m < Mc; m++) sum += val(m, 0);
            sums[0] = sum;
            for (int m = 0; m < Mc; m++)
                minheap_push(++heap_size, hv.data(), hi.data(), sum + (val(m, 1) - val(m, 0)), weights[m]);
            for (size_t kk = 1; kk < k; kk++) {
                const float s2 = sums[kk] = hv[0];
                const int64_t ti = terms[kk] = hi[0];
                do { minheap_pop(heap_size--, hv.data(), hi.data()); } while (heap_size > 0 && hi[0] == ti);
                int64_t ii = ti;
                for (int m = 0; m < Mc; m++) {
                    const int64_t nn = ii % kc;
                    ii /= kc;
                    if (nn + 1 >= kc) continue;
                    minheap_push(++heap_size, hv.data(), hi.data(),
                                 s2 + (val(m, (int)nn + 1) - val(m, (int)nn)), ti + weights[m]);
                }
            }
            for (size_t kk = 0; kk < k; kk++) {   // ranks -> centroid indices
                int64_t ii = terms[kk], ti = 0;
                for (int m = 0; m < Mc; m++) {
                    const int64_t nn = ii % kc;
                    ti += weights[m] * perm[m][nn];
                    ii /= kc;
                }
                terms[kk] = ti;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Product-quantizer tables (ProductQuantizer.cpp:410-436) for dsub < 16 (the
// non-BLAS branch; BASELINE configs have dsub = 8 or 6).
// ---------------------------------------------------------------------------
void orc_compute_inner_prod_table(const orc_index* ix, const float* x, float* tab) {
    for (int m = 0; m < ix->M; m++)
        for (int j = 0; j < ix->ksub; j++)
            tab[m * ix->ksub + j] = orc_fvec_inner_product(
                x + m * ix->dsub, ix->pq_centroids + ((size_t)m * ix->ksub + j) * ix->dsub, ix->dsub);
}

void orc_compute_distance_table(const orc_index* ix, const float* x, float* tab) {
    for (int m = 0; m < ix->M; m++)
        for (int j = 0; j < ix->ksub; j++)
            tab[m * ix->ksub + j] = orc_fvec_L2sqr(
                x + m * ix->dsub, ix->pq_centroids + ((size_t)m * ix->ksub + j) * ix->dsub, ix->dsub);
}

// IndexIVFPQ::precompute_table, use_precomputed_table == 1 (IndexIVFPQ.cpp:392-429):
// tab[i][m][j] = |cent_mj|^2 + 2 * <c_i|m, cent_mj>
void orc_precompute_table(const orc_index* ix, float* out) {
    const size_t mk = (size_t)ix->M * ix->ksub;
    std::vector<float> r_norms(mk);
    for (int m = 0; m < ix->M; m++)
        for (int j = 0; j < ix->ksub; j++)
            r_norms[m * ix->ksub + j] = orc_fvec_norm_L2sqr(
                ix->pq_centroids + ((size_t)m * ix->ksub + j) * ix->dsub, ix->dsub);
    if (ix->imi_nbits > 0) {
        // table type 2 (IndexIVFPQ.cpp:430-457): one row per coarse SUB-centroid index i, built
        // from the vector whose m-th part is the i-th centroid of coarse sub-quantizer m
        const int Mc = ix->imi_M, kc = 1 << ix->imi_nbits, dc = ix->d / Mc;
#pragma omp parallel for
        for (int i = 0; i < kc; i++) {
            std::vector<float> v(ix->d);
            for (int m = 0; m < Mc; m++)
                memcpy(&v[m * dc], ix->imi_centroids + ((size_t)m * kc + i) * dc, sizeof(float) * dc);
            float* tab = out + (size_t)i * mk;
            orc_compute_inner_prod_table(ix, v.data(), tab);
            fvec_madd(mk, r_norms.data(), 2.0f, tab, tab);
        }
        return;
    }
#pragma omp parallel for
    for (int i = 0; i < ix->nlist; i++) {
        float* tab = out + (size_t)i * mk;
        orc_compute_inner_prod_table(ix, ix->coarse_centroids + (size_t)i * ix->d, tab);
        fvec_madd(mk, r_norms.data(), 2.0f, tab, tab);
    }
}

// ---------------------------------------------------------------------------
// IndexIVFPQ::search_knn_with_key (IndexIVFPQ.cpp:964-1060) with
// InvertedListScanner::scan_list_with_table (:781-802) and
// QueryTables::init_query_L2 / precompute_list_tables_L2 (:557-563, :631-690).
// polysemous_ht == 0 and scan_table_threshold == 0 (the defaults, :57-62).
// Returns the number of codes visited (indexIVFPQ_stats.ncode, :1014,1035,1050).
// ---------------------------------------------------------------------------
int64_t orc_search_knn_with_key(const orc_index* ix, size_t nx, const float* qx,
                                const int64_t* keys, const float* coarse_dis,
                                size_t nprobe, size_t k, float* D, int64_t* I,
                                int store_pairs, int canonical) {
    const size_t mk = (size_t)ix->M * ix->ksub;
    const int d = ix->d;
    int64_t ncode_total = 0;
    int bad_key = 0;
#pragma omp parallel reduction(+ : ncode_total)
    {
        std::vector<float> sim_table(mk), sim_table_2(mk), residual(d);
        std::vector<Cand> cands;
#pragma omp for
        for (size_t i = 0; i < nx; i++) {
            const float* qi = qx + i * d;
            const int64_t* keysi = keys + i * nprobe;
            const float* cdi = coarse_dis + i * nprobe;
            float* heap_sim = D + i * k;
            int64_t* heap_ids = I + i * k;
            maxheap_heapify(k, heap_sim, heap_ids);
            cands.clear();

            // init_query_L2 (:557-563)
            if (!ix->by_residual) orc_compute_distance_table(ix, qi, sim_table.data());
            else if (ix->use_precomputed_table) orc_compute_inner_prod_table(ix, qi, sim_table_2.data());

            size_t nscan = 0;
            int64_t pos = 0;
            for (size_t ik = 0; ik < nprobe; ik++) {
                int64_t key = keysi[ik];
                if (key < 0) continue;                       // :1004-1007
                if (key >= ix->nlist) { bad_key = 1; continue; }  // reference throws (:1008-1011)
                size_t list_size = ix->list_offsets[key + 1] - ix->list_offsets[key];
                nscan += list_size;
                if (list_size == 0) continue;                // :1016

                // precompute_list_tables (:579-590, :631-690)
                float dis0 = 0;
                if (ix->by_residual) {
                    if (ix->use_precomputed_table == 0) {
                        const float* c = ix->coarse_centroids + (size_t)key * d;
                        for (int j = 0; j < d; j++) residual[j] = qi[j] - c[j];  // Index.cpp:76-81
                        orc_compute_distance_table(ix, residual.data(), sim_table.data());
                    } else if (ix->use_precomputed_table == 2) {
                        // IndexIVFPQ.cpp:645-686: per coarse sub-index ki the slice of Mf sub-quantizers
                        dis0 = cdi[ik];
                        const int Mf = ix->M / ix->imi_M;
                        int64_t kk = key;
                        for (int cm = 0; cm < ix->imi_M; cm++) {
                            const int64_t ki = kk & ((int64_t(1) << ix->imi_nbits) - 1);
                            kk >>= ix->imi_nbits;
                            const size_t o = (size_t)cm * Mf * ix->ksub;
                            fvec_madd((size_t)Mf * ix->ksub,
                                      ix->precomputed_table + ((size_t)ki * ix->M + (size_t)cm * Mf) * ix->ksub,
                                      -2.0f, sim_table_2.data() + o, sim_table.data() + o);
                        }
                    } else if (ix->float16_tables) {
                        // useFloat16LookupTables on the plain IVFPQ path, as the reference's GPU kernel forms the
                        // table (gpu/impl/PQScanMultiPassPrecomputed.cu:30-114 loadPrecomputedTerm with LookupT = half):
                        // term 2 and term 3 are kept as half (impl/IVFPQ.cu:599-684 / :1599-1680 toHalf), the table is
                        // their HALF sum, and the looked-up entries are accumulated in float (:431-449 ConvertTo<float>)
                        dis0 = cdi[ik];
                        const float* t2 = ix->precomputed_table + (size_t)key * mk;
                        for (size_t e = 0; e < mk; e++) {
                            const uint16_t h2 = (uint16_t)_cvtss_sh(t2[e], _MM_FROUND_TO_NEAREST_INT);
                            const uint16_t h3 = (uint16_t)_cvtss_sh(-2.0f * sim_table_2[e], _MM_FROUND_TO_NEAREST_INT);
                            // IEEE half add = the float sum of two halves rounded once more (24 >= 2*11 + 2 bits)
                            sim_table[e] = _cvtsh_ss((uint16_t)_cvtss_sh(_cvtsh_ss(h2) + _cvtsh_ss(h3), _MM_FROUND_TO_NEAREST_INT));
                        }
                    } else {
                        dis0 = cdi[ik];
                        fvec_madd(mk, ix->precomputed_table + (size_t)key * mk, -2.0f,
                                  sim_table_2.data(), sim_table.data());
                    }
                }
                // scan_list_with_table (:781-802)
                const uint8_t* lc = ix->codes + (size_t)ix->list_offsets[key] * ix->code_size;
                const int64_t* lids = ix->ids + ix->list_offsets[key];
                for (size_t j = 0; j < list_size; j++) {
                    float dis = dis0;
                    const float* tab = sim_table.data();
                    for (int m = 0; m < ix->M; m++) { dis += tab[*lc++]; tab += ix->ksub; }
                    int64_t id = store_pairs ? (key << 32 | (int64_t)j) : lids[j];
                    if (canonical) {
                        if (dis < FLT_MAX) cands.push_back({dis, pos + (int64_t)j, id});
                    } else if (dis < heap_sim[0]) {
                        maxheap_pop(k, heap_sim, heap_ids);
                        maxheap_push(k, heap_sim, heap_ids, dis, id);
                    }
                }
                pos += list_size;
                if (ix->max_codes && nscan >= (size_t)ix->max_codes) break;  // :1033
            }
            ncode_total += nscan;
            if (canonical) canonical_topk(cands, k, heap_sim, heap_ids);
            else maxheap_reorder(k, heap_sim, heap_ids);      // :1037
        }
    }
    return bad_key ? -1 : ncode_total;
}

// IndexIVFPQ::search (IndexIVFPQ.cpp:1063-1081)
int64_t orc_search(const orc_index* ix, size_t n, const float* x, size_t nprobe, size_t k,
                   float* D, int64_t* I, int canonical, int64_t* keys_out, float* cdis_out) {
    std::vector<int64_t> idx(n * nprobe);
    std::vector<float> cdis(n * nprobe);
    if (ix->imi_nbits > 0)
        orc_imi_search(ix, x, n, nprobe, cdis.data(), idx.data());
    else
        orc_knn_L2sqr(x, ix->coarse_centroids, ix->d, n, ix->nlist, nprobe, cdis.data(), idx.data(),
                      canonical, 0);
    if (keys_out) memcpy(keys_out, idx.data(), idx.size() * 8);
    if (cdis_out) memcpy(cdis_out, cdis.data(), cdis.size() * 4);
    return orc_search_knn_with_key(ix, n, x, idx.data(), cdis.data(), nprobe, k, D, I, 0, canonical);
}

// ---------------------------------------------------------------------------
// Add path (feeder): ProductQuantizer::compute_code (ProductQuantizer.cpp:311-336,
// first minimum wins, strict <) and IndexIVFPQ::add_core_o (IndexIVFPQ.cpp:192-272)
// minus the list append, which the caller does from (assign, codes).
// ---------------------------------------------------------------------------
void orc_pq_compute_codes(const orc_index* ix, const float* x, size_t n, uint8_t* codes) {
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        for (int m = 0; m < ix->M; m++) {
            float mindis = 1e20f;
            int idxm = -1;
            const float* xsub = x + i * ix->d + m * ix->dsub;
            for (int j = 0; j < ix->ksub; j++) {
                float dis = orc_fvec_L2sqr(
                    xsub, ix->pq_centroids + ((size_t)m * ix->ksub + j) * ix->dsub, ix->dsub);
                if (dis < mindis) { mindis = dis; idxm = j; }
            }
            codes[i * ix->code_size + m] = (uint8_t)idxm;
        }
    }
}

// assign[n] (1-NN coarse key) and codes[n][code_size] for n vectors.
// canonical selects the tie rule of the coarse 1-NN (see orc_knn_L2sqr).
void orc_encode(const orc_index* ix, const float* x, size_t n, int64_t* assign, uint8_t* codes,
                int canonical) {
    std::vector<float> dis(n);
    if (ix->imi_nbits > 0) orc_imi_search(ix, x, n, 1, dis.data(), assign);
    else orc_knn_L2sqr(x, ix->coarse_centroids, ix->d, n, ix->nlist, 1, dis.data(), assign, canonical, 0);
    if (ix->by_residual) {
        std::vector<float> res(n * (size_t)ix->d);
#pragma omp parallel for
        for (size_t i = 0; i < n; i++) {
            if (assign[i] < 0) { memset(&res[i * ix->d], 0, sizeof(float) * ix->d); continue; }
            if (ix->imi_nbits > 0) {   // MultiIndexQuantizer::reconstruct (IndexPQ.cpp:860-885)
                const int Mc = ix->imi_M, kc = 1 << ix->imi_nbits, dc = ix->d / Mc;
                int64_t jj = assign[i];
                for (int m = 0; m < Mc; m++) {
                    const float* c = ix->imi_centroids + ((size_t)m * kc + (jj % kc)) * dc;
                    jj /= kc;
                    for (int j = 0; j < dc; j++) res[i * ix->d + m * dc + j] = x[i * ix->d + m * dc + j] - c[j];
                }
                continue;
            }
            const float* c = ix->coarse_centroids + (size_t)assign[i] * ix->d;
            for (int j = 0; j < ix->d; j++) res[i * ix->d + j] = x[i * ix->d + j] - c[j];
        }
        orc_pq_compute_codes(ix, res.data(), n, codes);
    } else {
        orc_pq_compute_codes(ix, x, n, codes);
    }
}

// Replays a sequence of pushes through the heap: unit fixture for Heap.h semantics.
void orc_heap_topk(const float* vals, const int64_t* ids, size_t n, size_t k, float* D, int64_t* I) {
    maxheap_heapify(k, D, I);
    for (size_t j = 0; j < n; j++)
        if (vals[j] < D[0]) { maxheap_pop(k, D, I); maxheap_push(k, D, I, vals[j], ids[j]); }
    maxheap_reorder(k, D, I);
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
