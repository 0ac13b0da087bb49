// faiss::IDSelector and its two stock implementations (AuxIndexStructures.h:86-129): what remove_ids() takes.
#pragma once
#include <unordered_set>

#include "Index.h"

namespace faiss {

struct IDSelector {
  typedef Index::idx_t idx_t;
  virtual bool is_member(idx_t id) const = 0;
  virtual ~IDSelector() {}
};

/// ids in [imin, imax)
struct IDSelectorRange : IDSelector {
  idx_t imin, imax;
  IDSelectorRange(idx_t imin, idx_t imax) : imin(imin), imax(imax) {}
  bool is_member(idx_t id) const override { return id >= imin && id < imax; }
};

/// an explicit set of ids (the reference pairs a hash set with a Bloom filter for speed; membership is the same)
struct IDSelectorBatch : IDSelector {
  std::unordered_set<idx_t> set;
  IDSelectorBatch(long n, const idx_t* indices) : set(indices, indices + n) {}
  bool is_member(idx_t id) const override { return set.count(id) != 0; }
};

}  // namespace faiss
