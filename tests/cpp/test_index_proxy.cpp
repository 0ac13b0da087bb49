// CPU test of faiss::gpu::IndexProxy's host logic (include/faiss_amd/gpu/IndexProxy.h) with stub
// replicas: slices of ceil(n / #replicas) queries (gpu/IndexProxy.cpp:139-149), every replica writes
// its slice of the caller's buffers from its own thread, fan-out of train / add / reset, exception
// propagation, removeIndex, own_fields.  The GPU replicas are exercised in test_ivfpq_indexing.cpp.
#include <atomic>
#include <cstdio>
#include <set>
#include <thread>
#include <vector>

#include "faiss_amd/gpu/IndexProxy.h"

#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

static std::atomic<int> g_deleted{0};

struct Stub : faiss::Index {
  int tag;
  mutable faiss::Index::idx_t last_n = -1;
  mutable const float* last_x = nullptr;
  mutable std::thread::id tid;
  int adds = 0, trains = 0, resets = 0;
  bool fail_search = false;
  Stub(int d, int tag) : faiss::Index(d, faiss::METRIC_L2), tag(tag) {}
  ~Stub() override { g_deleted++; }
  void train(idx_t, const float*) override { trains++; is_trained = true; }
  void add(idx_t n, const float*) override { adds++; ntotal += n; }
  void reset() override { resets++; ntotal = 0; }
  void search(idx_t n, const float* x, idx_t k, float* D, idx_t* I) const override {
    if (fail_search) FAISS_THROW_MSG("stub failure");
    last_n = n; last_x = x; tid = std::this_thread::get_id();
    for (idx_t i = 0; i < n; i++)
      for (idx_t j = 0; j < k; j++) {
        D[i * k + j] = x[i * d] + (float)j;               // depends on the query only ...
        I[i * k + j] = (idx_t)(x[i * d] * 10) + j;
      }
  }
  void reconstruct(idx_t key, float* v) const override { for (int j = 0; j < d; j++) v[j] = (float)(key + tag); }
};

int main() {
  int failures = 0;
  const int d = 4, k = 3;
  for (int nrep : {1, 2, 3, 8}) {
    for (long n : {1L, 7L, 8L, 9L, 1000L, 1001L}) {
      faiss::gpu::IndexProxy proxy;
      std::vector<Stub*> reps;
      for (int r = 0; r < nrep; r++) { reps.push_back(new Stub(d, r)); proxy.addIndex(reps.back()); }
      proxy.own_fields = true;
      EXPECT(proxy.count() == nrep && proxy.d == d && proxy.at(0) == reps[0]);
      std::vector<float> x((size_t)n * d);
      for (long i = 0; i < n; i++) x[(size_t)i * d] = (float)i;
      std::vector<float> D((size_t)n * k, -1.f);
      std::vector<faiss::Index::idx_t> I((size_t)n * k, -1);
      proxy.search(n, x.data(), k, D.data(), I.data());
      const long per = (n + nrep - 1) / nrep;
      std::set<std::thread::id> tids;
      long covered = 0;
      for (int r = 0; r < nrep; r++) {
        faiss::Index::idx_t base, cnt;
        faiss::gpu::IndexProxy::sliceOf(n, nrep, r, &base, &cnt);
        EXPECT(base == std::min<long>(n, r * per) && cnt == std::max<long>(0, std::min<long>(per, n - r * per)));
        if (r * per < n) {
          EXPECT(reps[r]->last_n == cnt && reps[r]->last_x == x.data() + base * d);
          tids.insert(reps[r]->tid);
          covered += cnt;
        } else {
          EXPECT(reps[r]->last_n == -1);           // idle replica (IndexProxy.cpp:141-143)
        }
      }
      EXPECT(covered == n);
      EXPECT(tids.size() == (size_t)std::min<long>(nrep, (n + per - 1) / per));   // one thread per busy replica
      EXPECT(!tids.count(std::this_thread::get_id()));
      for (long i = 0; i < n; i++)
        for (int j = 0; j < k; j++) EXPECT(D[(size_t)i * k + j] == (float)i + j && I[(size_t)i * k + j] == i * 10 + j);
    }
  }
  {
    const int before = g_deleted.load();
    faiss::gpu::IndexProxy* proxy = new faiss::gpu::IndexProxy();
    Stub* a = new Stub(d, 0); Stub* b = new Stub(d, 1);
    proxy->addIndex(a); proxy->addIndex(b);
    std::vector<float> x(10 * d, 1.f);
    proxy->train(10, x.data());
    proxy->add(10, x.data());
    EXPECT(a->trains == 1 && b->trains == 1 && a->adds == 1 && b->adds == 1 && proxy->ntotal == 10 && proxy->is_trained);
    proxy->reset();
    EXPECT(a->resets == 1 && b->resets == 1 && proxy->ntotal == 0);
    float v[4];
    proxy->reconstruct(5, v);
    EXPECT(v[0] == 5.f);                              // from the first replica
    // a replica's exception reaches the caller
    b->fail_search = true;
    std::vector<float> D(10 * k);
    std::vector<faiss::Index::idx_t> I(10 * k);
    bool threw = false;
    try { proxy->search(10, x.data(), k, D.data(), I.data()); } catch (const faiss::FaissException&) { threw = true; }
    EXPECT(threw);
    // mismatching replica refused
    Stub* c = new Stub(d + 1, 2);
    threw = false;
    try { proxy->addIndex(c); } catch (const faiss::FaissException&) { threw = true; }
    EXPECT(threw && proxy->count() == 2);
    delete c;
    proxy->removeIndex(b);
    EXPECT(proxy->count() == 1);
    delete b;
    threw = false;
    try { proxy->removeIndex(b); } catch (const faiss::FaissException&) { threw = true; }
    EXPECT(threw);
    proxy->own_fields = true;
    delete proxy;                                    // deletes a
    EXPECT(g_deleted.load() - before == 3);
  }
  printf(failures ? "IndexProxy host test: %d FAILURES\n" : "IndexProxy host test: all ok\n", failures);
  return failures ? 1 : 0;
}
