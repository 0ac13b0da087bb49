// faiss::Index -- same public surface as the reference's Index.h:60-188.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <typeinfo>
#include <vector>

#include "FaissException.h"

namespace faiss {

enum MetricType { METRIC_INNER_PRODUCT = 0, METRIC_L2 = 1 };

struct IDSelector;
struct RangeSearchResult;

struct Index {
  typedef long idx_t;

  int d;
  idx_t ntotal;
  bool verbose;
  bool is_trained;
  MetricType metric_type;

  explicit Index(idx_t d = 0, MetricType metric = METRIC_INNER_PRODUCT)
      : d(d), ntotal(0), verbose(false), is_trained(true), metric_type(metric) {}
  virtual ~Index() {}

  virtual void train(idx_t /*n*/, const float* /*x*/) {}
  virtual void add(idx_t n, const float* x) = 0;
  virtual void add_with_ids(idx_t, const float*, const long*) {
    FAISS_THROW_MSG("add_with_ids not implemented for this type of index");
  }
  virtual void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const = 0;
  virtual void range_search(idx_t, const float*, float, RangeSearchResult*) const {
    FAISS_THROW_MSG("range search not implemented");
  }
  /// labels of the k nearest vectors (Index.cpp:23-29)
  void assign(idx_t n, const float* x, idx_t* labels, idx_t k = 1) {
    std::vector<float> distances((size_t)n * k);
    search(n, x, k, distances.data(), labels);
  }
  virtual void reset() = 0;
  virtual long remove_ids(const IDSelector&) { FAISS_THROW_MSG("remove_ids not implemented for this type of index"); }
  virtual void reconstruct(idx_t, float*) const { FAISS_THROW_MSG("reconstruct not implemented for this type of index"); }
  virtual void reconstruct_n(idx_t i0, idx_t ni, float* recons) const {
    for (idx_t i = 0; i < ni; i++) reconstruct(i0 + i, recons + i * d);
  }
  /// search, then reconstruct every result (Index.h:167, Index.cpp:57-74): the fork declares it right behind
  /// reconstruct_n, so it takes the same vtable slot here; a missing result (label -1) is filled with 0xff bytes
  virtual void search_and_reconstruct(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels,
                                      float* recons) const {
    search(n, x, k, distances, labels);
    for (idx_t i = 0; i < n; ++i)
      for (idx_t j = 0; j < k; ++j) {
        const idx_t ij = i * k + j, key = labels[ij];
        float* r = recons + ij * d;
        if (key < 0) memset(r, -1, sizeof(*r) * d);
        else reconstruct(key, r);
      }
  }
  /// residual = x - reconstruct(key) (Index.cpp:76-81)
  void compute_residual(const float* x, float* residual, idx_t key) const {
    reconstruct(key, residual);
    for (int i = 0; i < d; i++) residual[i] = x[i] - residual[i];
  }
  void display() const { printf("Index: %s  -> %ld elements\n", typeid(*this).name(), ntotal); }
};

}  // namespace faiss
