// faiss::MultiIndexQuantizer (IndexPQ.h:124-160, IndexPQ.cpp:781-900): the inverted
// multi-index coarse quantizer of the SIFT1B / Deep1B drivers ("IMI2x14").  Virtual
// index of ksub^M cells; search() = distance tables + the MinSumK walk, on the device
// (vlq_ivfpq_coarse_search of an index whose coarse quantizer is this codebook).
// Two sub-quantizers only, as in every driver of the reference.
#pragma once
#include "Index.h"
#include "ProductQuantizer.h"

namespace faiss {

struct MultiIndexQuantizer : Index {
  ProductQuantizer pq;
  int device = 0;

  MultiIndexQuantizer(int d, size_t M, size_t nbits) : Index(d, METRIC_L2), pq(d, M, nbits) {
    FAISS_THROW_IF_NOT_MSG(M == 2, "only 2-way multi-indexes are built");
    is_trained = false;
  }
  ~MultiIndexQuantizer() override { if (h_) vlq_ivfpq_destroy(h_); }
  MultiIndexQuantizer(const MultiIndexQuantizer&) = delete;
  MultiIndexQuantizer& operator=(const MultiIndexQuantizer&) = delete;

  void train(idx_t n, const float* x) override {
    pq.train((int)n, x);
    is_trained = true;
    ntotal = 1;
    for (size_t m = 0; m < pq.M; m++) ntotal *= (idx_t)pq.ksub;   // count of virtual elements
    dirty_ = true;
  }
  void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override {
    if (n == 0) return;
    FAISS_THROW_IF_NOT(is_trained);
    if (!h_ || dirty_) {
      if (h_) { vlq_ivfpq_destroy(h_); h_ = nullptr; }
      VLQ_CHECK(vlq_ivfpq_create(&h_, device, d, (int)ntotal, 2, 1));
      VLQ_CHECK(vlq_ivfpq_set_imi_centroids(h_, (int)pq.nbits, pq.centroids.data()));
      dirty_ = false;
    }
    VLQ_CHECK(vlq_ivfpq_coarse_search(h_, n, x, (int)k, distances, (int64_t*)labels));
  }
  /// concatenation of the sub-centroids of the cell (IndexPQ.cpp:860-885)
  void reconstruct(idx_t key, float* recons) const override {
    long jj = key;
    for (size_t m = 0; m < pq.M; m++) {
      memcpy(recons + m * pq.dsub, pq.get_centroids(m, jj % (long)pq.ksub), sizeof(float) * pq.dsub);
      jj /= (long)pq.ksub;
    }
  }
  void add(idx_t, const float*) override { FAISS_THROW_MSG("This index has virtual elements, it does not support add"); }
  void reset() override { FAISS_THROW_MSG("This index has virtual elements, it does not support reset"); }

 private:
  mutable vlq_ivfpq_t h_ = nullptr;
  mutable bool dirty_ = true;
};

}  // namespace faiss
