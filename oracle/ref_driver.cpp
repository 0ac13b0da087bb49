// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// Fixture generator: drives the *reference's own* CPU implementation (compiled
// from /root/reference by oracle/ref.mk into oracle/_ref/libfaiss_ref.so)
// through its public API and dumps inputs + outputs as a tagged binary that
// tests/golden/make_golden.py turns into the committed .npz fixtures.
//
// This file is ours: it only *calls* faiss::IndexFlatL2 / faiss::IndexIVFPQ
// (Index.h:60-188, IndexIVFPQ.h:29-164).  The reference cannot travel to the GPU
// box; the fixtures (data) do.
//
// usage: ref_driver <in.bin> <out.bin> [index_file]   (index_file: faiss::write_index of the index)
//        ref_driver bench <index_file> <in.bin> <out.bin> <nprobe> <k> <reps> <threads>
//            CPU baseline of bench.py: the reference's own read_index + IndexIVFPQ::search on the
//            host cores (in.bin: xq[nq*d]; out.bin: D, I of the last run + seconds per run)
//   in.bin : tagged arrays  cfg[int64 x 16], xt[nt*d], xb[nb*d], xq[nq*d],
//            optional xids[nb]
//   cfg = {d, nlist, M, nbits, nt, nb, nq, nprobe, k, max_codes, n_small,
//          kmeans_niter, pq_niter, by_residual, use_precomputed_table(-1=auto),
//          imi_nbits (0 = IndexFlatL2 coarse quantizer, else MultiIndexQuantizer 2 x imi_nbits)}

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "IndexFlat.h"
#include "IndexIVFPQ.h"
#include "IndexPQ.h"
#include "index_io.h"
#include "utils.h"

#include <omp.h>

#include <algorithm>
#include <chrono>

namespace {

struct Arr {
    char dtype;  // 'f' float32, 'l' int64, 'B' uint8
    std::vector<uint64_t> dims;
    std::vector<uint8_t> data;
};

size_t dsize(char t) { return t == 'f' ? 4 : t == 'l' ? 8 : 1; }

std::map<std::string, Arr> read_tagged(const char* fn) {
    std::map<std::string, Arr> m;
    FILE* f = fopen(fn, "rb");
    if (!f) { perror(fn); exit(1); }
    for (;;) {
        uint32_t nl;
        if (fread(&nl, 4, 1, f) != 1) break;
        std::string name(nl, ' ');
        if (fread(&name[0], 1, nl, f) != nl) exit(2);
        Arr a;
        uint32_t nd;
        if (fread(&a.dtype, 1, 1, f) != 1 || fread(&nd, 4, 1, f) != 1) exit(2);
        a.dims.resize(nd);
        size_t n = 1;
        for (uint32_t i = 0; i < nd; i++) {
            if (fread(&a.dims[i], 8, 1, f) != 1) exit(2);
            n *= a.dims[i];
        }
        a.data.resize(n * dsize(a.dtype));
        if (n && fread(a.data.data(), dsize(a.dtype), n, f) != n) exit(2);
        m[name] = a;
    }
    fclose(f);
    return m;
}

FILE* g_out;

void put(const char* name, char dtype, std::vector<uint64_t> dims, const void* p) {
    uint32_t nl = strlen(name), nd = dims.size();
    fwrite(&nl, 4, 1, g_out);
    fwrite(name, 1, nl, g_out);
    fwrite(&dtype, 1, 1, g_out);
    fwrite(&nd, 4, 1, g_out);
    size_t n = 1;
    for (auto d : dims) { fwrite(&d, 8, 1, g_out); n *= d; }
    if (n) fwrite(p, dsize(dtype), n, g_out);
}

}  // namespace

static int bench_main(int argc, char** argv) {
    if (argc != 9) { fprintf(stderr, "usage: %s bench index_file in.bin out.bin nprobe k reps threads\n", argv[0]); return 1; }
    const long nprobe = atol(argv[5]), k = atol(argv[6]), reps = atol(argv[7]), threads = atol(argv[8]);
    omp_set_num_threads((int)threads);
    faiss::Index* idx = faiss::read_index(argv[2]);          // index_io.cpp:459-532 (precomputes the table)
    faiss::IndexIVFPQ* index = dynamic_cast<faiss::IndexIVFPQ*>(idx);
    if (!index) { fprintf(stderr, "not an IndexIVFPQ file\n"); return 1; }
    auto in = read_tagged(argv[3]);
    const long nq = (long)in["xq"].dims[0];
    const float* xq = (const float*)in["xq"].data.data();
    index->nprobe = nprobe;
    std::vector<long> I(nq * k);
    std::vector<float> D(nq * k);
    index->search(nq, xq, k, D.data(), I.data());             // warm-up
    std::vector<double> secs;
    faiss::indexIVFPQ_stats.reset();
    for (long r = 0; r < reps; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        index->search(nq, xq, k, D.data(), I.data());
        secs.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    g_out = fopen(argv[4], "wb");
    if (!g_out) { perror(argv[4]); return 1; }
    put("D", 'f', {(uint64_t)nq, (uint64_t)k}, D.data());
    put("I", 'l', {(uint64_t)nq, (uint64_t)k}, I.data());
    std::vector<float> sf(secs.begin(), secs.end());
    put("seconds", 'f', {(uint64_t)reps}, sf.data());
    int64_t meta[3] = {index->use_precomputed_table, (int64_t)(faiss::indexIVFPQ_stats.ncode / reps), omp_get_max_threads()};
    put("meta", 'l', {3}, meta);
    fclose(g_out);
    delete idx;
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "bench") == 0) return bench_main(argc, argv);
    if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s in.bin out.bin [index_file]\n", argv[0]); return 1; }
    auto in = read_tagged(argv[1]);
    const int64_t* cfg = (const int64_t*)in["cfg"].data.data();
    const long d = cfg[0], nlist = cfg[1], M = cfg[2], nbits = cfg[3], nt = cfg[4],
               nb = cfg[5], nq = cfg[6], nprobe = cfg[7], k = cfg[8],
               max_codes = cfg[9], n_small = cfg[10], km_niter = cfg[11],
               pq_niter = cfg[12], by_residual = cfg[13], upt = cfg[14], imi_nbits = cfg[15];
    const float* xt = (const float*)in["xt"].data.data();
    const float* xb = (const float*)in["xb"].data.data();
    const float* xq = (const float*)in["xq"].data.data();
    const long* xids = in.count("xids") ? (const long*)in["xids"].data.data() : nullptr;

    g_out = fopen(argv[2], "wb");
    if (!g_out) { perror(argv[2]); return 1; }

    faiss::IndexFlatL2 coarse(d);
    faiss::MultiIndexQuantizer miq(d, 2, imi_nbits > 0 ? imi_nbits : 1);   // tests/sift1b_imi_pq.cpp:225-236
    faiss::Index* quant = imi_nbits > 0 ? (faiss::Index*)&miq : (faiss::Index*)&coarse;
    faiss::IndexIVFPQ index(quant, d, nlist, M, nbits);
    if (imi_nbits > 0) {
        index.quantizer_trains_alone = true;
        if (km_niter > 0) miq.pq.cp.niter = km_niter;
    }
    if (km_niter > 0) index.cp.niter = km_niter;
    if (pq_niter > 0) index.pq.cp.niter = pq_niter;
    index.by_residual = by_residual != 0;
    index.verbose = false;
    index.train(nt, xt);
    if (upt >= 0 && upt != index.use_precomputed_table) {
        index.use_precomputed_table = upt;   // user override, as the header allows
        if (upt == 0) index.precomputed_table.clear();
    }
    index.add_with_ids(nb, xb, xids);
    index.nprobe = nprobe;
    index.max_codes = max_codes;

    int64_t meta[4] = {index.use_precomputed_table, (int64_t)index.code_size,
                       (int64_t)index.pq.ksub, (int64_t)index.pq.dsub};
    put("meta", 'l', {4}, meta);
    if (imi_nbits > 0)
        put("imi_centroids", 'f', {2, miq.pq.ksub, miq.pq.dsub}, miq.pq.centroids.data());
    else
        put("coarse_centroids", 'f', {(uint64_t)nlist, (uint64_t)d}, coarse.xb.data());
    put("pq_centroids", 'f', {(uint64_t)M, index.pq.ksub, index.pq.dsub},
        index.pq.centroids.data());
    if (!index.precomputed_table.empty())
        put("precomputed_table", 'f',
            {(uint64_t)(index.precomputed_table.size() / (M * index.pq.ksub)), (uint64_t)M, index.pq.ksub},
            index.precomputed_table.data());

    // list-contiguous dump of the inverted lists (IndexIVF.h:55, IndexIVFPQ.h:43)
    std::vector<int64_t> off(nlist + 1, 0);
    for (long i = 0; i < nlist; i++) off[i + 1] = off[i] + index.ids[i].size();
    std::vector<uint8_t> codes(off[nlist] * index.code_size);
    std::vector<int64_t> ids(off[nlist]);
    for (long i = 0; i < nlist; i++) {
        if (index.ids[i].empty()) continue;
        memcpy(&codes[off[i] * index.code_size], index.codes[i].data(), index.codes[i].size());
        memcpy(&ids[off[i]], index.ids[i].data(), index.ids[i].size() * 8);
    }
    put("list_offsets", 'l', {(uint64_t)nlist + 1}, off.data());
    put("codes", 'B', {(uint64_t)off[nlist], index.code_size}, codes.data());
    put("ids", 'l', {(uint64_t)off[nlist]}, ids.data());

    // add path: coarse assignment of the database vectors (Index::assign)
    {
        std::vector<long> assign(nb);
        quant->assign(nb, xb, assign.data());
        put("xb_assign", 'l', {(uint64_t)nb}, assign.data());
    }

    // coarse stage (IndexIVFPQ.cpp:1073): blas path for nq >= 20
    std::vector<long> keys(nq * nprobe);
    std::vector<float> cdis(nq * nprobe);
    quant->search(nq, xq, nprobe, cdis.data(), keys.data());
    put("keys", 'l', {(uint64_t)nq, (uint64_t)nprobe}, keys.data());
    put("coarse_dis", 'f', {(uint64_t)nq, (uint64_t)nprobe}, cdis.data());

    // full search()
    std::vector<long> I(nq * k);
    std::vector<float> D(nq * k);
    faiss::indexIVFPQ_stats.reset();
    index.search(nq, xq, k, D.data(), I.data());
    int64_t ncode = faiss::indexIVFPQ_stats.ncode;
    put("D", 'f', {(uint64_t)nq, (uint64_t)k}, D.data());
    put("I", 'l', {(uint64_t)nq, (uint64_t)k}, I.data());
    put("ncode", 'l', {1}, &ncode);

    // the parity seam (IndexIVFPQ.h:140-146) with store_pairs=true
    {
        std::vector<long> Ip(nq * k);
        std::vector<float> Dp(nq * k);
        faiss::float_maxheap_array_t res = {size_t(nq), size_t(k), Ip.data(), Dp.data()};
        index.search_knn_with_key(nq, xq, keys.data(), cdis.data(), &res, true);
        put("D_pairs", 'f', {(uint64_t)nq, (uint64_t)k}, Dp.data());
        put("I_pairs", 'l', {(uint64_t)nq, (uint64_t)k}, Ip.data());
    }

    // small batch (< 20 queries): coarse stage takes the SSE path (utils.cpp:935-946)
    if (n_small > 0) {
        std::vector<long> ks(n_small * nprobe), Is(n_small * k);
        std::vector<float> cs(n_small * nprobe), Ds(n_small * k);
        quant->search(n_small, xq, nprobe, cs.data(), ks.data());
        index.search(n_small, xq, k, Ds.data(), Is.data());
        put("small_keys", 'l', {(uint64_t)n_small, (uint64_t)nprobe}, ks.data());
        put("small_coarse_dis", 'f', {(uint64_t)n_small, (uint64_t)nprobe}, cs.data());
        put("small_D", 'f', {(uint64_t)n_small, (uint64_t)k}, Ds.data());
        put("small_I", 'l', {(uint64_t)n_small, (uint64_t)k}, Is.data());
    }

    // per-query table part (ProductQuantizer.cpp:424-436 / :410-422) for a few queries
    {
        long nqt = nq < 8 ? nq : 8;
        std::vector<float> t(nqt * M * index.pq.ksub), t2(nqt * M * index.pq.ksub);
        for (long i = 0; i < nqt; i++) {
            index.pq.compute_inner_prod_table(xq + i * d, &t[i * M * index.pq.ksub]);
            index.pq.compute_distance_table(xq + i * d, &t2[i * M * index.pq.ksub]);
        }
        put("ip_table", 'f', {(uint64_t)nqt, (uint64_t)M, index.pq.ksub}, t.data());
        put("dis_table", 'f', {(uint64_t)nqt, (uint64_t)M, index.pq.ksub}, t2.data());
    }

    // norms (utils.cpp:675-682)
    {
        std::vector<float> qn(nq), cn(nlist);
        faiss::fvec_norms_L2sqr(qn.data(), xq, d, nq);
        put("q_norms", 'f', {(uint64_t)nq}, qn.data());
        if (imi_nbits == 0) {
            faiss::fvec_norms_L2sqr(cn.data(), coarse.xb.data(), d, nlist);
            put("c_norms", 'f', {(uint64_t)nlist}, cn.data());
        }
    }

    if (argc == 4) faiss::write_index(&index, argv[3]);   // index_io.cpp:240-355
    fclose(g_out);
    fprintf(stderr, "ref_driver: use_precomputed_table=%d ncode=%ld ntotal=%ld\n",
            index.use_precomputed_table, (long)ncode, (long)index.ntotal);
    return 0;
}
