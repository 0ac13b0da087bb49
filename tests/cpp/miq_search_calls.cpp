// the two coarse-stage entry points the interposer serves, called the way a program of the
// reference's users calls them, compiled against the REFERENCE's headers and linked like the reference's own drivers
// (interposer in front of the reference's library):
//   * faiss::MultiIndexQuantizer::search(n, x, k > 1, ...)  -- a caller of its own (IndexIVFPQR::search does this) --
//     lands in vlq_ivfpq_coarse_search of the handle that holds the quantizer's sub-centroids;
//   * faiss::IndexIVFPQ::search with nprobe beyond 1024 is served whole (coarse stage + scan in runs);
//   * faiss::IndexIVFPQ::search over a flat quantizer (BASELINE configs[1]'s class) is served whole too.
// Each result is compared with the reference's own definition of the same member, reached through dlsym on the reference's
// library: sub-vectors of 8 dimensions take the reference's SSE path (fvec_L2sqr, no BLAS) and must agree bit for bit --
// cells, sums, neighbours, distances; sub-vectors of 16 and more go through the BLAS vendor's sgemm there, so cells agree
// to rounding (>= 99.9 % equal, sums to 1e-5 relative).
//     usage: miq_search_calls <d: 16 | 32> ; prints one summary line, exit code 0 / 1
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "IndexFlat.h"
#include "IndexIVFPQ.h"
#include "IndexPQ.h"

typedef faiss::Index::idx_t idx_t;

int main(int argc, char** argv) {
    const int d = argc > 1 ? atoi(argv[1]) : 16;
    const size_t nbits = 6, kc = size_t(1) << nbits, nlist = kc * kc, M = 8;
    const size_t nt = 20000, nb = 60000, nq = 400;
    std::mt19937 rng(5);
    std::normal_distribution<float> gauss(0.f, 1.f);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    std::vector<float> centres(50 * d);
    for (auto& v : centres) v = uni(rng);
    auto gen = [&](size_t n) {
        std::vector<float> x(n * d);
        for (size_t i = 0; i < n; i++) {
            const size_t c = rng() % 50;
            for (int j = 0; j < d; j++) x[i * d + j] = centres[c * d + j] + 0.08f * gauss(rng);
        }
        return x;
    };
    std::vector<float> xt = gen(nt), xb = gen(nb), xq = gen(nq);

    faiss::MultiIndexQuantizer mq(d, 2, nbits);
    faiss::IndexIVFPQ index(&mq, d, nlist, M, 8);
    index.quantizer_trains_alone = true;
    index.verbose = false;
    index.train(nt, xt.data());
    index.add(nb, xb.data());
    index.precompute_table();

    void* ref = dlopen("libfaiss_ref.so", RTLD_NOW | RTLD_LOCAL);
    if (!ref) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    typedef void (*miq_fn)(const faiss::MultiIndexQuantizer*, idx_t, const float*, idx_t, float*, idx_t*);
    miq_fn miq_ref = (miq_fn)dlsym(ref, "_ZNK5faiss19MultiIndexQuantizer6searchElPKflPfPl");
    // (the reference's IndexIVFPQ::search body would reach the interposed members through its virtual calls: the CPU run of a
    // whole search is its two stages called one by one)
    typedef void (*knn_fn)(const faiss::IndexIVFPQ*, size_t, const float*, const long*, const float*, faiss::float_maxheap_array_t*, bool);
    knn_fn knn_ref = (knn_fn)dlsym(ref, "_ZNK5faiss10IndexIVFPQ19search_knn_with_keyEmPKfPKlS2_PNS_9HeapArrayINS_4CMaxIflEEEEb");
    if (!miq_ref || !knn_ref) { fprintf(stderr, "dlsym failed\n"); return 2; }

    const bool exact = d / 2 < 16;        // the reference's tables without BLAS
    int bad = 0;
    // 1. the quantizer alone, k > 1 (after the index has been synchronised: add() above went through the interposer)
    for (idx_t k : {2, 64, 300, 2048}) {
        std::vector<float> D1(nq * k), D0(nq * k);
        std::vector<idx_t> I1(nq * k), I0(nq * k);
        mq.search(nq, xq.data(), k, D1.data(), I1.data());
        miq_ref(&mq, nq, xq.data(), k, D0.data(), I0.data());
        size_t same = 0;
        double relmax = 0;
        for (size_t i = 0; i < nq * (size_t)k; i++) {
            same += I1[i] == I0[i];
            relmax = std::max(relmax, (double)std::fabs(D1[i] - D0[i]) / std::max(1e-12, (double)std::fabs(D0[i])));
        }
        const bool bits = memcmp(D1.data(), D0.data(), D0.size() * 4) == 0;
        const double frac = (double)same / (double)(nq * k);
        const bool ok = exact ? (same == nq * (size_t)k && bits) : (frac >= 0.999 && relmax <= 1e-5);
        printf("quantizer k=%ld: cells equal %.5f, sums rel. err %.2e%s -> %s\n", (long)k, frac, relmax, bits ? " (bit-equal)" : "", ok ? "ok" : "BAD");
        bad += !ok;
    }
    // 2. the whole search with more probes than one scan launch takes
    for (size_t nprobe : {size_t(32), size_t(1500), size_t(2048)}) {
        const idx_t k = 20;
        index.nprobe = nprobe;
        std::vector<float> D1(nq * k), D0(nq * k);
        std::vector<idx_t> I1(nq * k), I0(nq * k);
        index.search(nq, xq.data(), k, D1.data(), I1.data());
        std::vector<float> cdis(nq * nprobe);
        std::vector<long> keys(nq * nprobe);
        miq_ref(&mq, nq, xq.data(), (idx_t)nprobe, cdis.data(), keys.data());
        faiss::float_maxheap_array_t res = {nq, (size_t)k, I0.data(), D0.data()};
        knn_ref(&index, nq, xq.data(), keys.data(), cdis.data(), &res, false);
        size_t same = 0;
        double relmax = 0;
        for (size_t i = 0; i < nq * (size_t)k; i++) {
            same += I1[i] == I0[i];
            if (D0[i] < 1e30f) relmax = std::max(relmax, (double)std::fabs(D1[i] - D0[i]) / std::max(1e-12, (double)std::fabs(D0[i])));
        }
        const double frac = (double)same / (double)(nq * k);
        const bool ok = exact ? (frac >= 0.9999 && relmax == 0) : (frac >= 0.995 && relmax <= 1e-4);
        printf("whole search nprobe=%zu: neighbours equal %.5f, distances rel. err %.2e -> %s\n", nprobe, frac, relmax, ok ? "ok" : "BAD");
        bad += !ok;
    }
    // 3. the same whole search over a FLAT quantizer (IndexIVFPQ(IndexFlatL2, ...): BASELINE configs[1]'s class): the reference's
    // coarse stage is IndexFlat::search -> knn_L2sqr_blas for 20 and more queries (utils.cpp:834-901), whose sgemm order is the BLAS
    // vendor's, so probes at the nprobe-th rank may swap with a near-tie: neighbours to >= 99.5 %, distances to 1e-4; with fewer
    // than 20 queries both sides run fvec_L2sqr (utils.cpp:757-786) and must agree bit for bit
    {
        const size_t nl = 256;
        faiss::IndexFlatL2 fq(d);
        faiss::IndexIVFPQ fidx(&fq, d, nl, M, 8);
        fidx.verbose = false;
        fidx.train(nt, xt.data());
        fidx.add(nb, xb.data());
        fidx.precompute_table();
        fidx.nprobe = 16;
        typedef void (*flat_fn)(const faiss::IndexFlat*, idx_t, const float*, idx_t, float*, idx_t*);
        flat_fn flat_ref = (flat_fn)dlsym(ref, "_ZNK5faiss9IndexFlat6searchElPKflPfPl");
        if (!flat_ref) { fprintf(stderr, "dlsym IndexFlat::search failed\n"); return 2; }
        for (size_t n : {size_t(7), nq}) {
            const idx_t k = 20;
            std::vector<float> D1(n * k), D0(n * k), cdis(n * fidx.nprobe);
            std::vector<idx_t> I1(n * k), I0(n * k);
            std::vector<long> keys(n * fidx.nprobe);
            fidx.search(n, xq.data(), k, D1.data(), I1.data());
            flat_ref(&fq, n, xq.data(), (idx_t)fidx.nprobe, cdis.data(), keys.data());
            faiss::float_maxheap_array_t res = {n, (size_t)k, I0.data(), D0.data()};
            knn_ref(&fidx, n, xq.data(), keys.data(), cdis.data(), &res, false);
            size_t same = 0;
            double relmax = 0;
            for (size_t i = 0; i < n * (size_t)k; i++) {
                same += I1[i] == I0[i];
                if (I1[i] == I0[i] && D0[i] < 1e30f) relmax = std::max(relmax, (double)std::fabs(D1[i] - D0[i]) / std::max(1e-12, (double)std::fabs(D0[i])));
            }
            const double frac = (double)same / (double)(n * k);
            const bool ok = n < 20 ? (same == n * (size_t)k && memcmp(D1.data(), D0.data(), D0.size() * 4) == 0) : (frac >= 0.995 && relmax <= 1e-4);
            printf("flat quantizer, whole search, %zu queries: neighbours equal %.5f, distances rel. err %.2e -> %s\n", n, frac, relmax, ok ? "ok" : "BAD");
            bad += !ok;
        }
    }
    printf("miq_search_calls d=%d: %s\n", d, bad ? "FAILED" : "PASSED");
    return bad ? 1 : 0;
}
