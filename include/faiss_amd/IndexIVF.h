// faiss::IndexIVF (IndexIVF.h:45-108): inverted-file base with the reference's
// public fields (drivers poke nprobe / quantizer_trains_alone / cp directly).
#pragma once
#include <vector>

#include "Clustering.h"
#include "Heap.h"
#include "Index.h"

namespace faiss {

struct IndexIVF : Index {
  size_t nlist;
  size_t nprobe;
  Index* quantizer;
  bool quantizer_trains_alone;
  bool own_fields;
  ClusteringParameters cp;
  std::vector<std::vector<long> > ids;
  bool maintain_direct_map;
  std::vector<long> direct_map;

  IndexIVF(Index* quantizer, size_t d, size_t nlist, MetricType metric = METRIC_INNER_PRODUCT)
      : Index(d, metric), nlist(nlist), nprobe(1), quantizer(quantizer), quantizer_trains_alone(false),
        own_fields(false), ids(nlist), maintain_direct_map(false) {
    FAISS_THROW_IF_NOT(d == (size_t)quantizer->d);
    is_trained = quantizer->is_trained && (quantizer->ntotal == (idx_t)nlist);
    cp.niter = 10;   // IndexIVF.cpp:51
  }
  IndexIVF() : nlist(0), nprobe(1), quantizer(nullptr), quantizer_trains_alone(false), own_fields(false),
               maintain_direct_map(false) {}
  ~IndexIVF() override { if (own_fields) delete quantizer; }

  void reset() override {
    ntotal = 0;
    direct_map.clear();
    for (auto& l : ids) l.clear();
  }
  /// IndexIVF::train (IndexIVF.cpp:102-129)
  void train(idx_t n, const float* x) override {
    if (quantizer->is_trained && quantizer->ntotal == (idx_t)nlist) {
      if (verbose) printf("IVF quantizer does not need training.\n");
    } else if (quantizer_trains_alone) {
      quantizer->train(n, x);
      FAISS_THROW_IF_NOT_MSG(quantizer->ntotal == (idx_t)nlist, "nlist not consistent with quantizer size");
    } else {
      if (verbose) printf("Training IVF quantizer on %ld vectors in %dD\n", n, d);
      Clustering clus(d, (int)nlist, cp);
      quantizer->reset();
      clus.train(n, x, *quantizer);
      quantizer->is_trained = true;
    }
    train_residual(n, x);
    is_trained = true;
  }
  void add(idx_t n, const float* x) override { add_with_ids(n, x, nullptr); }
  virtual void train_residual(idx_t, const float*) {}
  /// IndexIVF::merge_from (IndexIVF.cpp:150-177): moves the other index's lists into this one, `add_id` added to its ids
  virtual void merge_from_residuals(IndexIVF&) { FAISS_THROW_MSG("merge_from_residuals not implemented for this type of index"); }
  void merge_from(IndexIVF& other, idx_t add_id) {
    FAISS_THROW_IF_NOT(other.d == d && other.nlist == nlist);
    FAISS_THROW_IF_NOT_MSG(!maintain_direct_map && !other.maintain_direct_map, "direct map copy not implemented");
    FAISS_THROW_IF_NOT_MSG(typeid(*this) == typeid(other), "can only merge indexes of the same type");
    for (size_t i = 0; i < nlist; i++) {
      std::vector<long>& src = other.ids[i];
      std::vector<long>& dest = ids[i];
      for (size_t j = 0; j < src.size(); j++) dest.push_back(src[j] + add_id);
      src.clear();
    }
    merge_from_residuals(other);
    ntotal += other.ntotal;
    other.ntotal = 0;
  }
  size_t get_list_size(size_t list_no) const { return ids[list_no].size(); }
  /// 1 = perfectly balanced (IndexIVF.cpp:140-147)
  double imbalance_factor() const {
    double tot = 0, uf = 0;
    for (size_t i = 0; i < nlist; i++) { tot += ids[i].size(); uf += ids[i].size() * (double)ids[i].size(); }
    return tot > 0 ? uf * nlist / (tot * tot) : 0;
  }
};

}  // namespace faiss
