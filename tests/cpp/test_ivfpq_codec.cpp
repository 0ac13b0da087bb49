// The reference's tests/test_ivfpq_codec.cpp:27-65 replayed against the faiss-compatible shell on an MI355X (part 1:
// same sizes, same drand48 stream, same two monotonicity expectations; encode_multiple runs on the device), then the
// members of faiss::IndexIVFPQ / faiss::Index that surround the hot path (IndexIVFPQ.h:75-162, Index.h:167): each is
// held to a host recomputation of what the reference's implementation does (IndexIVFPQ.cpp, file:line at each part).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "faiss_amd/AuxIndexStructures.h"
#include "faiss_amd/IndexFlat.h"
#include "faiss_amd/IndexIVFPQ.h"

#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)
static int failures = 0;

static const int d = 64;
static const size_t nb = 8000;

static double l2sqr(const float* a, const float* b, size_t n) {
  double s = 0;
  for (size_t i = 0; i < n; i++) { const double t = (double)a[i] - b[i]; s += t * t; }
  return s;
}

// tests/test_ivfpq_codec.cpp:27-48
static double eval_codec_error(long ncentroids, long m, const std::vector<float>& v) {
  faiss::IndexFlatL2 coarse_quantizer(d);
  faiss::IndexIVFPQ index(&coarse_quantizer, d, ncentroids, m, 8);
  index.pq.cp.niter = 10;   // speed up train
  index.train(nb, v.data());
  std::vector<long> keys(nb);
  std::vector<uint8_t> codes(nb * m);
  index.encode_multiple(nb, keys.data(), v.data(), codes.data(), true);
  std::vector<float> v2(nb * d);
  index.decode_multiple(nb, keys.data(), codes.data(), v2.data());
  // encode (one vector, host) must agree with the device's encode_multiple, and compute_keys = false with = true
  std::vector<uint8_t> one(m), again(nb * m);
  for (size_t i = 0; i < nb; i += 7) {   // (every 7th: the host encoder is one thread)
    index.encode(keys[i], &v[i * d], one.data());
    EXPECT(memcmp(one.data(), &codes[i * m], m) == 0);
  }
  index.encode_multiple(nb, keys.data(), v.data(), again.data(), false);
  EXPECT(again == codes);
  return l2sqr(v.data(), v2.data(), nb * d);
}

int main() {
  srand48(0);   // (the reference's test does not seed: drand48's default state = seed 0 of this libc)
  std::vector<float> database(nb * d);
  for (auto& x : database) x = drand48();

  // part 1: TEST(IVFPQ, codec), tests/test_ivfpq_codec.cpp:52-65
  const double err0 = eval_codec_error(16, 8, database);
  const double err1 = eval_codec_error(128, 8, database);   // more coarse centroids: more accurate
  const double err2 = eval_codec_error(16, 16, database);   // more PQ bytes: more accurate
  printf("part 1 (codec): err0=%.1f err1=%.1f err2=%.1f\n", err0, err1, err2);
  EXPECT(err0 > err1);
  EXPECT(err0 > err2);

  // a small populated index for the remaining parts
  const size_t n = 3000;
  const int nlist = 20, M = 8;
  faiss::IndexFlatL2 cq(d);
  faiss::IndexIVFPQ index(&cq, d, nlist, M, 8);
  index.pq.cp.niter = 8;
  index.maintain_direct_map = true;
  index.train(n, database.data());
  // duplicates on purpose: vectors 100..104 are copies of vector 7
  std::vector<float> xb(database.begin(), database.begin() + n * d);
  for (size_t i = 100; i < 105; i++) memcpy(&xb[i * d], &xb[7 * d], sizeof(float) * d);
  index.add(n, xb.data());

  // part 2: reconstruct_n (IndexIVFPQ.cpp:278-303) == decode_multiple of the stored codes; reconstruct (:306-322)
  {
    std::vector<float> rec(200 * (size_t)d), one(d);
    index.reconstruct_n(50, 200, rec.data());
    int bad = 0;
    for (int i = 0; i < 200; i += 13) {
      index.reconstruct(50 + i, one.data());
      for (int j = 0; j < d; j++) if (fabsf(one[j] - rec[(size_t)i * d + j]) > 1e-6f) bad++;
    }
    EXPECT(bad == 0);
    EXPECT(l2sqr(rec.data(), &xb[50 * (size_t)d], 200 * (size_t)d) < 0.35 * 200 * d / 12.0 * 4);   // a lossy code, not garbage
    printf("part 2 (reconstruct_n): ok, error %.2f per vector\n", l2sqr(rec.data(), &xb[50 * (size_t)d], 200 * (size_t)d) / 200);
  }

  // part 3: Index::search_and_reconstruct (Index.h:167, Index.cpp:57-74): results of search + their reconstructions,
  // missing results filled with 0xff bytes
  {
    const int nq = 20, k = 4;
    index.nprobe = 3;
    std::vector<float> D((size_t)nq * k), D2((size_t)nq * k), R((size_t)nq * k * d), one(d);
    std::vector<faiss::Index::idx_t> I((size_t)nq * k), I2((size_t)nq * k);
    const faiss::Index& base = index;           // through the base class: the virtual's slot
    base.search_and_reconstruct(nq, xb.data(), k, D.data(), I.data(), R.data());
    index.search(nq, xb.data(), k, D2.data(), I2.data());
    EXPECT(D == D2 && I == I2);
    int bad = 0;
    for (int i = 0; i < nq * k; i++) {
      if (I[i] < 0) { unsigned char ff[4]; memcpy(ff, &R[(size_t)i * d], 4); if (ff[0] != 0xff) bad++; continue; }
      index.reconstruct(I[i], one.data());
      if (memcmp(one.data(), &R[(size_t)i * d], sizeof(float) * d) != 0) bad++;
    }
    EXPECT(bad == 0);
    printf("part 3 (search_and_reconstruct): ok\n");
  }

  // part 4: find_duplicates (IndexIVFPQ.cpp:1239-1280): the copies of vector 7 form one group
  {
    std::vector<faiss::Index::idx_t> ids(index.ntotal);
    std::vector<size_t> lims(index.ntotal / 2 + 2);
    const size_t ng = index.find_duplicates(ids.data(), lims.data());
    bool found = false;
    for (size_t g = 0; g < ng; g++) {
      std::vector<long> grp(ids.begin() + lims[g], ids.begin() + lims[g + 1]);
      std::sort(grp.begin(), grp.end());
      EXPECT(grp.size() >= 2);
      if (std::find(grp.begin(), grp.end(), 7) != grp.end()) {
        found = true;
        for (long id = 100; id < 105; id++) EXPECT(std::find(grp.begin(), grp.end(), id) != grp.end());
      }
    }
    EXPECT(found);
    printf("part 4 (find_duplicates): %zu groups\n", ng);
  }

  // part 5: copy_subset_to (IndexIVFPQ.cpp:337-361) + merge_from (IndexIVF.cpp:150-177 -> merge_from_residuals :327-335):
  // two halves by id range, merged back, search like the original
  {
    faiss::IndexIVFPQ a(&cq, d, nlist, M, 8), b(&cq, d, nlist, M, 8);
    for (faiss::IndexIVFPQ* x : {&a, &b}) {
      x->pq = index.pq; x->is_trained = true; x->nprobe = 4; x->use_precomputed_table = 0;
      x->precompute_table();
    }
    index.copy_subset_to(a, 0, 0, 1700);
    index.copy_subset_to(b, 0, 1700, (long)n);
    EXPECT(a.ntotal == 1700 && b.ntotal == (long)n - 1700);
    faiss::IndexIVFPQ none(&cq, d, nlist, M, 8);
    index.copy_subset_to(none, 1, 2, 0);                      // type 1: the reference's loop copies nothing
    EXPECT(none.ntotal == 0);
    // ids of b shifted down by 1700 and back up by merge_from's add_id
    for (auto& l : b.ids) for (auto& id : l) id -= 1700;
    a.merge_from(b, 1700);
    EXPECT(a.ntotal == (long)n && b.ntotal == 0);
    const int nq = 50, k = 5;
    index.nprobe = 4;
    std::vector<float> D1((size_t)nq * k), D2((size_t)nq * k);
    std::vector<faiss::Index::idx_t> I1((size_t)nq * k), I2((size_t)nq * k);
    index.search(nq, &database[5000 * (size_t)d], k, D1.data(), I1.data());
    a.search(nq, &database[5000 * (size_t)d], k, D2.data(), I2.data());
    EXPECT(D1 == D2);                                         // same codes in every list, in another order: same distances
    int same = 0;
    for (int i = 0; i < nq * k; i++) same += I1[i] == I2[i];
    EXPECT(same >= nq * k * 0.95);                            // (ties between duplicates may come back in another order)
    printf("part 5 (copy_subset_to + merge_from): distances equal, %d of %d labels equal\n", same, nq * k);
  }

  // part 6: remove_ids (IndexIVFPQ.cpp:1195-1224) + the device copy follows
  {
    faiss::IndexIVFPQ c(&cq, d, nlist, M, 8);
    c.pq = index.pq; c.is_trained = true; c.nprobe = nlist; c.precompute_table();
    index.copy_subset_to(c, 0, 0, (long)n);
    faiss::IDSelectorRange sel(100, 105);
    std::vector<float> D(8);
    std::vector<faiss::Index::idx_t> I(8);
    c.search(1, &xb[7 * (size_t)d], 8, D.data(), I.data());
    int copies = 0;
    for (auto id : I) copies += (id >= 100 && id < 105);
    EXPECT(copies == 5);
    EXPECT(c.remove_ids(sel) == 5 && c.ntotal == (long)n - 5);
    c.search(1, &xb[7 * (size_t)d], 8, D.data(), I.data());
    copies = 0;
    for (auto id : I) copies += (id >= 100 && id < 105);
    EXPECT(copies == 0 && I[0] == 7);
    const long some[3] = {7, 9, 11};
    faiss::IDSelectorBatch batch(3, some);
    EXPECT(c.remove_ids(batch) == 3);
    printf("part 6 (remove_ids): ok\n");
  }

  // part 7: train_residual_o's second-level residuals (IndexIVFPQ.cpp:73-132) = training residual minus its PQ decode,
  // and the default constructor (IndexIVFPQ.cpp:1227-1236)
  {
    faiss::IndexFlatL2 q2(d);
    faiss::IndexIVFPQ t(&q2, d, 16, 8, 8);
    t.pq.cp.niter = 5;
    const size_t nt = 2000;
    faiss::Clustering clus(d, 16, t.cp);
    clus.train(nt, database.data(), q2);
    std::vector<float> r2(nt * d);
    t.train_residual_o(nt, database.data(), r2.data());
    std::vector<long> keys(nt);
    std::vector<uint8_t> codes(nt * 8);
    std::vector<float> dec(nt * d);
    t.is_trained = true;
    t.encode_multiple(nt, keys.data(), database.data(), codes.data(), true);
    t.decode_multiple(nt, keys.data(), codes.data(), dec.data());
    double worst = 0;
    for (size_t i = 0; i < nt * d; i++) worst = std::max(worst, (double)fabsf((database[i] - dec[i]) - r2[i]));
    EXPECT(worst < 1e-5);
    faiss::IndexIVFPQ def;
    EXPECT(def.use_precomputed_table == 0 && def.scan_table_threshold == 0 && def.polysemous_ht == 0 && def.max_codes == 0);
    printf("part 7 (train_residual_o, default ctor): ok (max deviation %.2g)\n", worst);
  }

  if (failures) { printf("%d FAILURES\n", failures); return 1; }
  printf("all ok\n");
  return 0;
}
