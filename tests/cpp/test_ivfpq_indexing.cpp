// C++ parity test of the faiss-compatible shell (include/faiss_amd) on an MI355X.
// Part 1 follows the reference's own unit test tests/test_ivfpq_indexing.cpp:20-100
// (same shapes, same drand48 stream, same acceptance bar n_ok > 0.4 * nq); part 2
// follows the shape of gpu/test/TestGpuIndexIVFPQ.cpp (GPU index built from a CPU
// index must answer like it -- here: identically); part 3 checks error behaviour.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "faiss_amd/IndexFlat.h"
#include "faiss_amd/IndexIVFPQ.h"
#include "faiss_amd/gpu/GpuClonerOptions.h"
#include "faiss_amd/gpu/GpuIndexIVFPQ.h"
#include "faiss_amd/gpu/IndexProxy.h"
#ifndef VLQ_NO_RCCL   // tests/cpp/Makefile: an image without librccl builds everything but part 2h
#include "faiss_amd/gpu/RcclShardedIndex.h"
#endif
#include "faiss_amd/gpu/StandardGpuResources.h"
#include "faiss_amd/index_io.h"

#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

int main() {
  int failures = 0;
  const int d = 64;
  const size_t nb = 1000, nt = 1500;
  const int ncentroids = 25, nq = 200, k = 5;

  faiss::IndexFlatL2 coarse_quantizer(d);
  faiss::IndexIVFPQ index(&coarse_quantizer, d, ncentroids, 16, 8);
  faiss::IndexFlatL2 index_gt(d);

  srand48(35);
  std::vector<float> trainvecs(nt * d), database(nb * d), queries((size_t)nq * d);
  for (auto& v : trainvecs) v = drand48();
  index.train(nt, trainvecs.data());
  for (auto& v : database) v = drand48();
  index.add(nb, database.data());
  index_gt.add(nb, database.data());
  for (auto& v : queries) v = drand48();

  std::vector<faiss::Index::idx_t> gt_nns(nq), nns((size_t)k * nq);
  std::vector<float> gt_dis(nq), dis((size_t)k * nq);
  index_gt.search(nq, queries.data(), 1, gt_dis.data(), gt_nns.data());
  index.nprobe = 5;
  index.search(nq, queries.data(), k, dis.data(), nns.data());
  int n_ok = 0;
  for (int q = 0; q < nq; q++)
    for (int i = 0; i < k; i++)
      if (nns[q * k + i] == gt_nns[q]) n_ok++;
  printf("part 1: n_ok = %d of %d (bar %d), ntotal=%ld, use_precomputed_table=%d, imbalance=%.2f\n", n_ok, nq,
         (int)(nq * 0.4), index.ntotal, index.use_precomputed_table, index.imbalance_factor());
  EXPECT(n_ok > nq * 0.4);
  EXPECT(index.use_precomputed_table == 1);                       // IndexIVFPQ.cpp:128-130,396-408
  EXPECT(index.precomputed_table.size() == (size_t)ncentroids * 16 * 256);
  EXPECT(faiss::indexIVFPQ_stats.nq == (size_t)nq && faiss::indexIVFPQ_stats.ncode > 0);

  // part 2: GPU index copied from the CPU-side object answers identically
  faiss::gpu::StandardGpuResources res;
  faiss::gpu::GpuIndexIVFPQConfig config;
  config.usePrecomputedTables = true;
  faiss::gpu::GpuIndexIVFPQ gpuIndex(&res, &index, config);
  gpuIndex.setNumProbes(5);
  std::vector<faiss::Index::idx_t> gnns((size_t)k * nq);
  std::vector<float> gdis((size_t)k * nq);
  gpuIndex.search(nq, queries.data(), k, gdis.data(), gnns.data());
  EXPECT(gnns == nns);
  EXPECT(gdis == dis);
  EXPECT(gpuIndex.ntotal == (faiss::Index::idx_t)nb && gpuIndex.getNumLists() == ncentroids);
  EXPECT(gpuIndex.getNumSubQuantizers() == 16 && gpuIndex.getBitsPerCode() == 8 &&
         gpuIndex.getCentroidsPerSubQuantizer() == 256);
  for (int l = 0; l < ncentroids; l++) {
    EXPECT(gpuIndex.getListLength(l) == (int)index.ids[l].size());
    EXPECT(gpuIndex.getListIndices(l) == index.ids[l]);
    EXPECT(gpuIndex.getListCodes(l) == index.codes[l]);
  }
  // copyTo round trip (TestGpuIndexIVFPQ.cpp CopyTo)
  faiss::IndexFlatL2 q2(d);
  faiss::IndexIVFPQ back(&q2, d, ncentroids, 16, 8);
  gpuIndex.copyTo(&back);
  back.nprobe = 5;
  std::vector<faiss::Index::idx_t> bnns((size_t)k * nq);
  std::vector<float> bdis((size_t)k * nq);
  back.search(nq, queries.data(), k, bdis.data(), bnns.data());
  EXPECT(bnns == nns && bdis == dis);
  // an empty GPU index trained and filled on the device
  faiss::gpu::GpuIndexIVFPQ fresh(&res, d, ncentroids, 16, 8, faiss::METRIC_L2, config);
  fresh.train(nt, trainvecs.data());
  fresh.add(nb, database.data());
  fresh.setNumProbes(5);
  fresh.search(nq, queries.data(), k, gdis.data(), gnns.data());
  EXPECT(gnns == nns && gdis == dis);       // same training recipe, same seeds -> same index

  // part 2b: the fork's VLQ index (shape of gpu/test/demo_ivfpq_line_indexing_gpu.cpp):
  // train / add / search on the device.  On this structure-free uniform data lines add
  // nothing over plain residuals, so the bar is a little under the IVFPQ bar of part 1.
  {
    faiss::gpu::GpuIndexIVFPQ vlqIndex(&res, d, ncentroids, 16, 8, /*nedge*/ 8, /*nLambda*/ 32, faiss::METRIC_L2, config);
    vlqIndex.train(nt, trainvecs.data());
    vlqIndex.add(nb, database.data());
    vlqIndex.setNumProbes(5);
    vlqIndex.w1_ = 40;   // every line of the 5 probed centroids
    std::vector<faiss::Index::idx_t> vnns((size_t)k * nq);
    std::vector<float> vdis((size_t)k * nq);
    vlqIndex.search(nq, queries.data(), k, vdis.data(), vnns.data());
    int v_ok = 0;
    for (int q = 0; q < nq; q++)
      for (int i = 0; i < k; i++)
        if (vnns[q * k + i] == gt_nns[q]) v_ok++;
    printf("part 2b (VLQ): n_ok = %d of %d (bar %d), ntotal=%ld, edge 0 of centroid 0 -> %d (len2 %.3f), lambda[0]=%.3f\n",
           v_ok, nq, (int)(nq * 0.3), vlqIndex.ntotal, vlqIndex.edgeInfo_[0], vlqIndex.edgeDistInfo_[0], vlqIndex.lambdaInfo_[0]);
    EXPECT(v_ok > nq * 0.3);
    EXPECT(vlqIndex.isVLQ() && vlqIndex.ntotal == (faiss::Index::idx_t)nb);
    {   // reset() empties the lines; adding again gives the same answers
      std::vector<faiss::Index::idx_t> n0((size_t)k * nq);
      std::vector<float> d0((size_t)k * nq);
      vlqIndex.search(nq, queries.data(), k, d0.data(), n0.data());
      vlqIndex.reset();
      EXPECT(vlqIndex.ntotal == 0);
      vlqIndex.add(nb, database.data());
      std::vector<faiss::Index::idx_t> n1((size_t)k * nq);
      std::vector<float> d1((size_t)k * nq);
      vlqIndex.search(nq, queries.data(), k, d1.data(), n1.data());
      EXPECT(n0 == n1 && d0 == d1);
    }
    // the fork's raw files: a second index loaded from .ppqt/.db* answers identically, and
    // two list-range shards (readDbFromFile(name, 2, r)) merged with merge() do too
    vlqIndex.writeCodebookToFile("/tmp/vlq_test");
    vlqIndex.writeDbToFile("/tmp/vlq_test");
    std::vector<faiss::Index::idx_t> parts_n((size_t)2 * nq * k);
    std::vector<float> parts_d((size_t)2 * nq * k);
    for (int r = -1; r < 2; r++) {
      faiss::gpu::GpuIndexIVFPQ re(&res, d, ncentroids, 16, 8, 8, 32, faiss::METRIC_L2, config);
      re.readCodebookFromFile("/tmp/vlq_test");
      if (r < 0) re.readDbFromFile("/tmp/vlq_test"); else re.readDbFromFile("/tmp/vlq_test", 2, r);
      re.setNumProbes(5);
      re.w1_ = 40;
      std::vector<faiss::Index::idx_t> rn((size_t)k * nq);
      std::vector<float> rd((size_t)k * nq);
      re.search(nq, queries.data(), k, rd.data(), rn.data());
      if (r < 0) { EXPECT(rn == vnns && rd == vdis); }
      else {
        std::copy(rn.begin(), rn.end(), parts_n.begin() + (size_t)r * nq * k);
        std::copy(rd.begin(), rd.end(), parts_d.begin() + (size_t)r * nq * k);
      }
    }
    std::vector<faiss::Index::idx_t> mn((size_t)k * nq);
    std::vector<float> md((size_t)k * nq);
    vlqIndex.merge(parts_n.data(), parts_d.data(), k, nq, 2, md.data(), mn.data());
    EXPECT(md == vdis);
    {   // the overloads the fork's MPI drivers call (gpu/test/deep1b16_query.cpp:270: readDbFromFile(prename, 0, numproces, rank)),
        // the reference's partition arithmetic (gpu/GpuIndexIVFPQ.cu:2127-2141: nl / pronum lines per rank, the remainder of a
        // non-dividing pronum dropped), getListLambdas (:2332-2338) and writeCentroidsToFile (:1760-1771)
      const int nl = ncentroids * 8;
      long total3 = 0, whole = 0;
      for (int i = 0; i < nl; i++) whole += vlqIndex.getListLength(i);
      for (int r = 0; r < 3; r++) {
        faiss::gpu::GpuIndexIVFPQ re(&res, d, ncentroids, 16, 8, 8, 32, faiss::METRIC_L2, config);
        re.readCodebookFromFile("/tmp/vlq_test");
        re.readDbFromFile("/tmp/vlq_test", 0, 3, r);
        const int per = nl / 3;
        long mine = 0;
        for (int i = 0; i < nl; i++) {
          const bool in = i >= per * r && i < per * (r + 1);
          EXPECT(re.getListLength(i) == (in ? vlqIndex.getListLength(i) : 0));
          if (in) mine += re.getListLength(i);
        }
        EXPECT(re.ntotal == mine);
        EXPECT(re.begin_ == (ncentroids / 3) * r && re.end_ == (r == 2 ? ncentroids - 1 : re.begin_ + ncentroids / 3 - 1));
        total3 += mine;
      }
      long tail = 0;
      for (int i = (nl / 3) * 3; i < nl; i++) tail += vlqIndex.getListLength(i);
      EXPECT(total3 + tail == whole);
      faiss::gpu::GpuIndexIVFPQ re(&res, d, ncentroids, 16, 8, 8, 32, faiss::METRIC_L2, config);
      re.readCodebookFromFile("/tmp/vlq_test");
      re.readDbFromFile("/tmp/vlq_test", (size_t)whole);
      EXPECT(re.ntotal == whole);
      for (int i = 0; i < nl; i += 7) {
        EXPECT(re.getListLambdas(i) == vlqIndex.getListLambdas(i));
        EXPECT(re.getListLambdas(i).size() == (size_t)re.getListLength(i));
        EXPECT(re.getListCodes(i).size() == (size_t)re.getListLength(i) * 16);
      }
      bool threw = false;
      try { re.readDbFromFile("/tmp/vlq_test", (size_t)1); } catch (const faiss::FaissException&) { threw = true; }
      EXPECT(threw);
      vlqIndex.writeCentroidsToFile("/tmp/vlq_test_centroids");
      std::ifstream cf("/tmp/vlq_test_centroids.umem", std::ifstream::binary);
      size_t num = 0, dim = 0;
      cf >> num >> dim;
      EXPECT(num == (size_t)ncentroids && dim == (size_t)d);
      cf.seekg(20, std::ios::beg);
      std::vector<float> cen((size_t)ncentroids * d);
      cf.read((char*)cen.data(), cen.size() * sizeof(float));
      EXPECT(cf.good());
      remove("/tmp/vlq_test_centroids.umem");
      printf("part 2d: readDbFromFile(name, nb, pronum, rank) / (name, nb), getListLambdas, writeCentroidsToFile ok\n");
    }
    for (const char* ext : {".ppqt", ".dbIdx", ".dblas", ".dbcodes", ".dbcount"}) remove((std::string("/tmp/vlq_test") + ext).c_str());
    // returned distances omit |q|^2 (gpu/impl/Distance.cu:286-291): adding it back gives >= 0
    for (int q = 0; q < 10; q++) {
      double qn = 0;
      for (int j = 0; j < d; j++) qn += (double)queries[q * d + j] * queries[q * d + j];
      EXPECT(vdis[q * k] + qn > -1e-3);
    }
  }

  // part 2e: the configuration block of the reference's VLQ drivers, as they write it
  // (gpu/test/deep1b16_query.cpp:224-243; gpu/test/sift1b16_query.cpp likewise): float16 look-up
  // tables requested through GpuClonerOptions, ids "on the CPU", a temp-memory fraction.  It must
  // construct, train, add and search.  The VLQ search then runs with float16 look-up tables (as the
  // reference does): its answers stay within the reference's own GPU-vs-CPU bar of the fp32 answers
  // (relative distance error <= 0.015, gpu/test/TestGpuIndexIVFPQ.cpp:89-99); so does the plain IVFPQ index
  // (16 x 8-bit codes: half tables as well).
  {
    faiss::gpu::StandardGpuResources resources;
    resources.setTempMemoryFraction(0.25);
    int dev_no = 0;
    faiss::gpu::GpuIndexIVFPQConfig cfg16;
    cfg16.device = dev_no;
    faiss::gpu::GpuClonerOptions co;
    co.useFloat16 = true;
    cfg16.indicesOptions = faiss::gpu::INDICES_CPU;
    cfg16.flatConfig.useFloat16 = co.useFloat16CoarseQuantizer;
    cfg16.flatConfig.storeTransposed = co.storeTransposed;
    cfg16.useFloat16LookupTables = co.useFloat16;
    cfg16.usePrecomputedTables = co.usePrecomputed;
    EXPECT(co.usePrecomputed && co.indicesOptions == faiss::gpu::INDICES_64_BIT && co.reserveVecs == 0);
    faiss::gpu::GpuIndexIVFPQ drv(&resources, d, ncentroids, 16, 8, /*nedge*/ 8, /*nLambda*/ 32, faiss::METRIC_L2, cfg16);
    faiss::gpu::GpuIndexIVFPQConfig cfg32;
    cfg32.usePrecomputedTables = true;
    faiss::gpu::GpuIndexIVFPQ base(&resources, d, ncentroids, 16, 8, 8, 32, faiss::METRIC_L2, cfg32);
    std::vector<faiss::Index::idx_t> n16((size_t)k * nq), n32((size_t)k * nq);
    std::vector<float> d16((size_t)k * nq), d32((size_t)k * nq);
    faiss::gpu::GpuIndexIVFPQ* both[2] = {&drv, &base};
    for (auto* ix : both) {
      ix->train(nt, trainvecs.data());
      ix->add(nb, database.data());
      ix->setNumProbes(5);
      ix->w1_ = 40;
    }
    drv.search(nq, queries.data(), k, d16.data(), n16.data());
    base.search(nq, queries.data(), k, d32.data(), n32.data());
    EXPECT(drv.getFloat16LookupTables() && !base.getFloat16LookupTables());
    {
      int same = 0;
      double worst = 0;
      for (size_t i = 0; i < n16.size(); i++) {
        same += n16[i] == n32[i];
        // returned distances omit |q|^2: compare on the scale of the true distance
        double qn = 0;
        for (int j = 0; j < d; j++) qn += (double)queries[(i / k) * d + j] * queries[(i / k) * d + j];
        const double rel = std::fabs((double)d16[i] - d32[i]) / std::max(1e-6, (double)d32[i] + qn);
        if (n16[i] == n32[i] && rel > worst) worst = rel;
      }
      printf("part 2e: fp16 tables vs fp32: %d of %zu labels equal, max relative distance error %.2e\n", same, n16.size(), worst);
      EXPECT(same >= (int)(0.9 * n16.size()));       // TestGpuIndexIVFPQ.cpp: <= 10 % differing results
      EXPECT(worst <= 0.015);
    }
    // the same options on the plain IVFPQ copy-constructor (GpuAutoTune.cpp's cloner fills them in alike)
    faiss::gpu::GpuIndexIVFPQ copy16(&resources, &index, cfg16);
    copy16.setNumProbes(5);
    copy16.search(nq, queries.data(), k, d16.data(), n16.data());
    {   // half tables on the plain path too (16 x 8-bit codes): the reference's GPU-vs-CPU bar against the fp32 answer
        // (gpu/test/TestGpuIndexIVFPQ.cpp:89-99: relative distance error <= 0.015, few results differing)
      int same = 0;
      double worst = 0;
      for (size_t i = 0; i < n16.size(); i++) {
        same += n16[i] == nns[i];
        if (n16[i] == nns[i]) worst = std::max(worst, std::fabs((double)d16[i] - dis[i]) / std::max(1e-6, (double)dis[i]));
      }
      printf("part 2e: plain IVFPQ, fp16 tables vs fp32: %d of %zu labels equal, max relative distance error %.2e\n", same, n16.size(), worst);
      EXPECT(same >= (int)(0.9 * n16.size()) && worst <= 0.015 && d16 != dis);
    }
    faiss::gpu::GpuMultipleClonerOptions mco;
    EXPECT(!mco.shard && mco.usePrecomputed);
    printf("part 2e: reference driver configuration (useFloat16LookupTables, INDICES_CPU) accepted\n");
  }

  // part 2f: faiss::gpu::IndexProxy (gpu/IndexProxy.cpp:123-168) over GPU replicas: one replica = the
  // index itself; two replicas (both on device 0 here: a one-GPU box) searched from two host threads at
  // once -- slices written straight into the caller's buffers, answers identical to one index
  {
    faiss::gpu::GpuIndexIVFPQ rep0(&res, &index, config), rep1(&res, &index, config);
    rep0.setNumProbes(5);
    rep1.setNumProbes(5);
    for (int nrep = 1; nrep <= 2; nrep++) {
      faiss::gpu::IndexProxy proxy;
      proxy.addIndex(&rep0);
      if (nrep == 2) proxy.addIndex(&rep1);
      std::vector<faiss::Index::idx_t> pn((size_t)k * nq, -7);
      std::vector<float> pd((size_t)k * nq, -7.f);
      for (int rep = 0; rep < 3; rep++) proxy.search(nq, queries.data(), k, pd.data(), pn.data());
      EXPECT(pn == nns && pd == dis);
      EXPECT(proxy.count() == nrep && proxy.ntotal == (faiss::Index::idx_t)nb && proxy.d == d);
      // fewer queries than replicas (the <20-query coarse path: compare with the replica itself)
      std::vector<faiss::Index::idx_t> one_n(k), one_p(k);
      std::vector<float> one_d(k), one_pd(k);
      proxy.search(1, queries.data(), k, one_pd.data(), one_p.data());
      rep0.search(1, queries.data(), k, one_d.data(), one_n.data());
      EXPECT(one_p == one_n && one_pd == one_d);
    }
    printf("part 2f: IndexProxy over 1 and 2 GPU replicas answers like one index\n");
  }

  // part 2g: GpuResources::getPinnedMemory (gpu/GpuResources.h:40, StandardGpuResources.cpp:24: one 256 MB
  // page-locked block): queries and results placed there take the asynchronous copy path and return the same rows
  {
    faiss::gpu::StandardGpuResources pres;
    std::pair<void*, size_t> pin = pres.getPinnedMemory();
    EXPECT(pin.first != nullptr && pin.second == ((size_t)256 << 20));
    EXPECT(pres.getPinnedMemory().first == pin.first);              // one block, handed out again
    hipPointerAttribute_t attr;
    EXPECT(hipPointerGetAttributes(&attr, pin.first) == hipSuccess && attr.type == hipMemoryTypeHost);
    float* pq = (float*)pin.first;
    float* pD = pq + (size_t)nq * d;
    faiss::Index::idx_t* pI = (faiss::Index::idx_t*)(pD + (size_t)nq * k);
    std::copy(queries.begin(), queries.end(), pq);
    faiss::gpu::GpuIndexIVFPQ gp(&pres, &index);
    gp.setNumProbes(5);
    gp.search(nq, pq, k, pD, pI);
    EXPECT(std::equal(nns.begin(), nns.end(), pI) && std::equal(dis.begin(), dis.end(), pD));
    printf("part 2g: %zu MB of pinned memory from GpuResources; search from / into it equals the pageable call\n", pin.second >> 20);
  }

  // part 2h: the RCCL host in C++ (north star: "host code stays C++ ... RCCL all-gather of per-shard top-k"):
  // replicas, ceil(n / G) slices, one grouped all-gather per output array.  One GPU on the box = one rank; the slice
  // arithmetic for G > 1 is checked on its own.
#ifndef VLQ_NO_RCCL
  {
    long lo, hi, per;
    faiss::gpu::RcclShardedIndex::sliceOf(10000, 8, 7, &lo, &hi, &per);
    EXPECT(per == 1250 && lo == 8750 && hi == 10000);
    faiss::gpu::RcclShardedIndex::sliceOf(10, 8, 5, &lo, &hi, &per);       // IndexProxy.cpp:139-149: short and empty slices
    EXPECT(per == 2 && lo == 10 && hi == 10);
    faiss::gpu::RcclShardedIndex::sliceOf(10, 8, 4, &lo, &hi, &per);
    EXPECT(lo == 8 && hi == 10);
    faiss::gpu::StandardGpuResources rres;
    faiss::gpu::GpuIndexIVFPQ rep(&rres, &index);
    rep.setNumProbes(5);
    faiss::gpu::RcclShardedIndex sharded({&rep});
    std::vector<faiss::Index::idx_t> sn((size_t)k * nq);
    std::vector<float> sd((size_t)k * nq);
    sharded.search(nq, queries.data(), k, sd.data(), sn.data());
    EXPECT(sn == nns && sd == dis);
    // a second, smaller batch through the same buffers (>= 20 queries: below that the reference's coarse path --
    // and ours -- is the SSE one, which rounds differently from the GEMM path, utils.cpp:935-946)
    sharded.search(40, queries.data(), k, sd.data(), sn.data());
    EXPECT(std::equal(sn.begin(), sn.begin() + 40 * k, nns.begin()) && std::equal(sd.begin(), sd.begin() + 40 * k, dis.begin()));
    printf("part 2h: RcclShardedIndex (%d rank): rows gathered over RCCL equal the single index\n", sharded.numReplicas());
  }
#endif

  // part 2c: inverted multi-index coarse quantizer (the "IMI2x.." indexes of
  // tests/sift1b_imi_pq.cpp:225-236): quantizer_trains_alone, table type 2
  {
    faiss::MultiIndexQuantizer miq(d, 2, 3);                       // 2 x 3 bit = 64 cells
    faiss::IndexIVFPQ imi(&miq, d, 64, 16, 8);
    imi.quantizer_trains_alone = true;
    imi.train(nt, trainvecs.data());
    imi.add(nb, database.data());
    imi.nprobe = 12;
    std::vector<faiss::Index::idx_t> inns((size_t)k * nq);
    std::vector<float> idis((size_t)k * nq);
    imi.search(nq, queries.data(), k, idis.data(), inns.data());
    int i_ok = 0;
    for (int q = 0; q < nq; q++)
      for (int i = 0; i < k; i++)
        if (inns[q * k + i] == gt_nns[q]) i_ok++;
    printf("part 2c (IMI): n_ok = %d of %d (bar %d), table type %d, table rows %zu\n", i_ok, nq, (int)(nq * 0.4),
           imi.use_precomputed_table, imi.precomputed_table.size() / (16 * 256));
    EXPECT(i_ok > nq * 0.4);
    EXPECT(imi.use_precomputed_table == 2 && imi.precomputed_table.size() == (size_t)8 * 16 * 256);
    EXPECT(miq.ntotal == 64);
  }

  // part 2d: write_index / read_index (the drivers cache *_populated_index.faissindex,
  // tests/sift1b_imi_pq.cpp:238-293): a reloaded index answers identically
  {
    const char* fn = "/tmp/vlq_test_index.faissindex";
    faiss::write_index(&index, fn);
    faiss::Index* loaded = faiss::read_index(fn);
    faiss::IndexIVFPQ* liv = dynamic_cast<faiss::IndexIVFPQ*>(loaded);
    EXPECT(liv != nullptr && liv->use_precomputed_table == 1 && liv->ntotal == index.ntotal);
    liv->nprobe = 5;
    std::vector<faiss::Index::idx_t> lnns((size_t)k * nq);
    std::vector<float> ldis((size_t)k * nq);
    loaded->search(nq, queries.data(), k, ldis.data(), lnns.data());
    EXPECT(lnns == nns && ldis == dis);
    delete loaded;
    remove(fn);
  }

  // part 3: error behaviour (FAISS_THROW_* -> FaissException)
  bool threw = false;
  try { faiss::IndexIVFPQ bad(&coarse_quantizer, d, ncentroids, 16, 9); } catch (const faiss::FaissException&) { threw = true; }
  EXPECT(threw);
  threw = false;
  try { gpuIndex.setNumProbes(5000); } catch (const faiss::FaissException&) { threw = true; }
  EXPECT(threw);
  threw = false;
  try {   // invalid key through the seam: reference prints "Invalid key" and throws
    std::vector<long> keys((size_t)nq * 5, 99);
    std::vector<float> cd((size_t)nq * 5, 0.f);
    faiss::float_maxheap_array_t r = {(size_t)nq, (size_t)k, nns.data(), dis.data()};
    index.search_knn_with_key(nq, queries.data(), keys.data(), cd.data(), &r);
  } catch (const faiss::FaissException&) { threw = true; }
  EXPECT(threw);

  printf(failures ? "C++ shell test: %d FAILURES\n" : "C++ shell test: all ok\n", failures);
  return failures ? 1 : 0;
}
