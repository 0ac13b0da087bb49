// faiss::IndexFlat / IndexFlatL2 (IndexFlat.h:23-87): stores the vectors, searches
// exhaustively.  On the hot path it is the coarse quantizer; its search() is the
// MFMA distance kernel + wave64 select behind vlq_ivfpq_coarse_search
// (= knn_L2sqr, utils.cpp:935-946).  L2 only: inner product is outside the path.
#pragma once
#include <vector>

#include "Index.h"

namespace faiss {

struct IndexFlat : Index {
  std::vector<float> xb;   ///< database vectors, ntotal * d

  explicit IndexFlat(idx_t d, MetricType metric = METRIC_INNER_PRODUCT) : Index(d, metric) {}
  IndexFlat() {}
  ~IndexFlat() override { if (h_) vlq_ivfpq_destroy(h_); }
  IndexFlat(const IndexFlat&) = delete;
  IndexFlat& operator=(const IndexFlat&) = delete;

  void add(idx_t n, const float* x) override {
    xb.insert(xb.end(), x, x + n * d);
    ntotal += n;
    dirty_ = true;
  }
  void reset() override { xb.clear(); ntotal = 0; dirty_ = true; }

  void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override {
    FAISS_THROW_IF_NOT_MSG(metric_type == METRIC_L2, "only METRIC_L2 is built on the device path");
    FAISS_THROW_IF_NOT_MSG(k >= 1 && k <= VLQ_MAX_NPROBE, "k outside 1..1024");
    if (n == 0) return;
    if (ntotal == 0) {   // heap_heapify + reorder of an empty heap (Heap.h:204-207,318-321)
      for (idx_t i = 0; i < n * k; i++) { distances[i] = 3.402823466e+38f; labels[i] = -1; }
      return;
    }
    sync_();
    VLQ_CHECK(vlq_ivfpq_coarse_search(h_, n, x, (int)k, distances, (int64_t*)labels));
  }
  void reconstruct(idx_t key, float* recons) const override {
    FAISS_THROW_IF_NOT(key >= 0 && key < ntotal);
    memcpy(recons, &xb[(size_t)key * d], sizeof(float) * d);
  }

  /// device on which search() runs (set before the first search)
  int device = 0;

 private:
  void sync_() const {
    if (!dirty_ && h_) return;
    if (h_) { vlq_ivfpq_destroy(h_); h_ = nullptr; }
    VLQ_CHECK(vlq_ivfpq_create(&h_, device, d, (int)ntotal, 1, 1));
    VLQ_CHECK(vlq_ivfpq_set_coarse_centroids(h_, xb.data()));
    dirty_ = false;
  }
  mutable vlq_ivfpq_t h_ = nullptr;
  mutable bool dirty_ = true;
};

struct IndexFlatL2 : IndexFlat {
  explicit IndexFlatL2(idx_t d) : IndexFlat(d, METRIC_L2) {}
  IndexFlatL2() {}
};

}  // namespace faiss
