// faiss::IndexIVFPQ (IndexIVFPQ.h:29-164) over the MI355X library: same public data
// model (ids / codes per list, pq, precomputed_table, search-time fields), add()
// and search() run on the device through the C ABI.  The host-side lists stay the
// authoritative copy -- exactly what copyFrom / write_index consumers read -- and are
// mirrored to HBM list-contiguously before the first search after a change.
#pragma once
#include <cstdint>
#include <vector>

#include <algorithm>

#include "AuxIndexStructures.h"
#include "IndexFlat.h"
#include "IndexIVF.h"
#include "IndexPQ.h"
#include "ProductQuantizer.h"

namespace faiss {

struct IndexIVFPQStats {   // IndexIVFPQ.h:169-195
  size_t nq, nlist, ncode, nrefine, n_hamming_pass;
  size_t assign_cycles, search_cycles, refine_cycles;
  size_t init_query_cycles, init_list_cycles, scan_cycles, heap_cycles;
  IndexIVFPQStats() { reset(); }
  void reset() { memset(this, 0, sizeof(*this)); }
};
inline IndexIVFPQStats indexIVFPQ_stats;   // "global var that collects them all" (IndexIVFPQ.h:195); C++17

struct IndexIVFPQ : IndexIVF {
  bool by_residual;
  int use_precomputed_table;
  size_t code_size;
  ProductQuantizer pq;
  bool do_polysemous_training;
  void* polysemous_training;       // unused: polysemous_ht != 0 is outside the path
  size_t scan_table_threshold;
  size_t max_codes;
  int polysemous_ht;
  std::vector<std::vector<uint8_t> > codes;
  std::vector<float> precomputed_table;

  /// device the index lives on (set before train/add/search)
  int device = 0;

  IndexIVFPQ(Index* quantizer, size_t d, size_t nlist, size_t M, size_t nbits_per_idx)
      : IndexIVF(quantizer, d, nlist, METRIC_L2), pq(d, M, nbits_per_idx) {
    FAISS_THROW_IF_NOT(nbits_per_idx <= 8);   // IndexIVFPQ.cpp:51
    code_size = pq.code_size;
    is_trained = false;
    codes.resize(nlist);
    by_residual = true;
    use_precomputed_table = 0;
    scan_table_threshold = 0;
    max_codes = 0;
    polysemous_training = nullptr;
    do_polysemous_training = false;
    polysemous_ht = 0;
  }
  /// IndexIVFPQ() (IndexIVFPQ.cpp:1227-1236): the run-time fields only; read_index / the caller fills in the rest
  IndexIVFPQ() : by_residual(true), use_precomputed_table(0), code_size(0), do_polysemous_training(false),
                 polysemous_training(nullptr), scan_table_threshold(0), max_codes(0), polysemous_ht(0) {}
  ~IndexIVFPQ() override { if (h_) vlq_ivfpq_destroy(h_); }
  IndexIVFPQ(const IndexIVFPQ&) = delete;
  IndexIVFPQ& operator=(const IndexIVFPQ&) = delete;

  void train_residual(idx_t n, const float* x) override { train_residual_o(n, x, nullptr); }

  /// train_residual_o (IndexIVFPQ.cpp:73-132); residuals_2 (optional, [n'][d] with n' = the possibly subsampled training
  /// set size): training vector minus its PQ reconstruction, what IVFPQR trains its refinement quantizer on
  void train_residual_o(idx_t n, const float* x, float* residuals_2) {
    size_t ns = n;
    std::vector<float> xs = maybe_subsample(d, &ns, pq.cp.max_points_per_centroid * pq.ksub, x, pq.cp.seed);
    std::vector<float> trainset;
    if (by_residual) {
      std::vector<idx_t> assign(ns);
      quantizer->assign(ns, xs.data(), assign.data());
      trainset.resize(ns * d);
      for (size_t i = 0; i < ns; i++) quantizer->compute_residual(&xs[i * d], &trainset[i * d], assign[i]);
    } else {
      trainset.swap(xs);
    }
    pq.verbose = verbose;
    pq.train((int)ns, trainset.data());
    FAISS_THROW_IF_NOT_MSG(!do_polysemous_training, "polysemous training is outside the built path");
    hdirty_ = true;
    if (residuals_2) {   // IndexIVFPQ.cpp:113-125: codes of the training set itself (no coarse step), decoded and subtracted
      std::vector<uint8_t> tc(ns * pq.code_size);
      std::vector<int64_t> zero(ns, 0);
      if (ns > 0) {   // pq.compute_codes of the training set as it stands: the device encoder without its residual step
        sync_(false);
        VLQ_CHECK(vlq_ivfpq_set_search_options(h_, 0, 0, (int64_t)max_codes));
        VLQ_CHECK(vlq_ivfpq_encode_preassigned(h_, (int64_t)ns, trainset.data(), zero.data(), tc.data()));
        sync_(false);                                // puts by_residual back
      }
      for (size_t i = 0; i < ns; i++) {
        float* res = residuals_2 + i * d;
        pq.decode(&tc[i * pq.code_size], res);
        for (int j = 0; j < d; j++) res[j] = trainset[i * d + j] - res[j];
      }
    }
    if (by_residual) precompute_table();
  }

  /// precompute_table (IndexIVFPQ.cpp:392-459), flat-L2 quantizer: table type 1,
  /// computed on the device and mirrored into `precomputed_table`
  void precompute_table() {
    const MultiIndexQuantizer* miq = dynamic_cast<const MultiIndexQuantizer*>(quantizer);
    if (use_precomputed_table == 0)      // choose the type of table (IndexIVFPQ.cpp:396-408)
      use_precomputed_table = (miq && pq.M % miq->pq.M == 0) ? 2 : 1;
    FAISS_THROW_IF_NOT_MSG((use_precomputed_table == 2) == (miq != nullptr), "table type does not match the quantizer");
    sync_(false);
    precomputed_table.resize((miq ? miq->pq.ksub : nlist) * pq.M * pq.ksub);
    VLQ_CHECK(vlq_ivfpq_get_precomputed_table(h_, precomputed_table.data()));
  }

  void add_with_ids(idx_t n, const float* x, const long* xids) override { add_core_o(n, x, xids, nullptr); }

  /// add_core_o (IndexIVFPQ.cpp:192-272): assignment + encoding on the device,
  /// append to the host lists in input order
  void add_core_o(idx_t n, const float* x, const long* xids, float* residuals_2,
                  const long* precomputed_idx = nullptr) {
    FAISS_THROW_IF_NOT(is_trained);
    FAISS_THROW_IF_NOT_MSG(!residuals_2 && !precomputed_idx, "IVFPQR / precomputed_idx are outside the built path");
    if (n == 0) return;
    sync_(false);
    std::vector<int64_t> idx(n);
    std::vector<uint8_t> xcodes((size_t)n * code_size);
    VLQ_CHECK(vlq_ivfpq_encode(h_, n, x, idx.data(), xcodes.data()));
    for (idx_t i = 0; i < n; i++) {
      const int64_t key = idx[i];
      if (key < 0) continue;
      ids[key].push_back(xids ? xids[i] : ntotal + i);
      codes[key].insert(codes[key].end(), &xcodes[i * code_size], &xcodes[(i + 1) * code_size]);
      if (maintain_direct_map) direct_map.push_back(key << 32 | (long)(ids[key].size() - 1));
    }
    ntotal += n;
    ldirty_ = true;
  }

  void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override {
    check_search_();
    sync_(true);
    VLQ_CHECK(vlq_ivfpq_search(h_, n, x, (int)nprobe, (int)k, distances, (int64_t*)labels));
    collect_stats_(n);
  }

  /// the parity seam (IndexIVFPQ.h:140-146)
  virtual void search_knn_with_key(size_t nx, const float* qx, const long* keys, const float* coarse_dis,
                                   float_maxheap_array_t* res, bool store_pairs = false) const {
    check_search_();
    sync_(true);
    VLQ_CHECK(vlq_ivfpq_search_preassigned(h_, (int64_t)nx, qx, (const int64_t*)keys, coarse_dis, (int)nprobe,
                                           (int)res->k, res->val, (int64_t*)res->ids, store_pairs ? 1 : 0));
    collect_stats_(nx);
  }

  void reset() override {
    IndexIVF::reset();
    for (auto& c : codes) c.clear();
    ldirty_ = true;
  }

  /// code of one vector for list `key` (IndexIVFPQ.cpp:136-144); host-side, for single vectors
  void encode(long key, const float* x, uint8_t* code) const {
    if (by_residual) {
      std::vector<float> r(d);
      quantizer->compute_residual(x, r.data(), key);
      pq.compute_code(r.data(), code);
    } else {
      pq.compute_code(x, code);
    }
  }

  /// encode_multiple (IndexIVFPQ.cpp:150-167) on the device: compute_keys = true also fills keys with the nearest list
  void encode_multiple(size_t n, long* keys, const float* x, uint8_t* xcodes, bool compute_keys = false) const {
    if (n == 0) return;
    sync_(false);
    if (compute_keys) VLQ_CHECK(vlq_ivfpq_encode(h_, (int64_t)n, x, (int64_t*)keys, xcodes));
    else VLQ_CHECK(vlq_ivfpq_encode_preassigned(h_, (int64_t)n, x, (const int64_t*)keys, xcodes));
  }

  /// inverse of encode_multiple (IndexIVFPQ.cpp:169-183)
  void decode_multiple(size_t n, const long* keys, const uint8_t* xcodes, float* x) const {
    pq.decode(xcodes, x, n);
    if (by_residual) {
      std::vector<float> centroid(d);
      for (size_t i = 0; i < n; i++) {
        quantizer->reconstruct(keys[i], centroid.data());
        float* xi = x + i * d;
        for (int j = 0; j < d; j++) xi[j] += centroid[j];
      }
    }
  }

  /// IndexIVFPQ.cpp:278-303: every stored vector whose id lies in [i0, i0 + ni)
  void reconstruct_n(idx_t i0, idx_t ni, float* recons) const override {
    FAISS_THROW_IF_NOT(ni == 0 || (i0 >= 0 && i0 + ni <= ntotal));
    std::vector<float> centroid(d);
    for (size_t key = 0; key < nlist; key++) {
      const std::vector<long>& idlist = ids[key];
      const uint8_t* code_line = codes[key].data();
      for (size_t ofs = 0; ofs < idlist.size(); ofs++) {
        const long id = idlist[ofs];
        if (!(id >= i0 && id < i0 + ni)) continue;
        float* r = recons + (size_t)d * (id - i0);
        pq.decode(code_line + ofs * pq.code_size, r);
        if (by_residual) {
          quantizer->reconstruct((idx_t)key, centroid.data());
          for (int j = 0; j < d; j++) r[j] += centroid[j];
        }
      }
    }
  }

  /// IndexIVFPQ.cpp:1195-1224: a removed entry is replaced by the list's last one
  long remove_ids(const IDSelector& sel) override {
    FAISS_THROW_IF_NOT_MSG(!maintain_direct_map, "direct map remove not implemented");
    long nremove = 0;
    for (size_t i = 0; i < nlist; i++) {
      std::vector<long>& idsi = ids[i];
      uint8_t* codesi = codes[i].data();
      long l = (long)idsi.size(), j = 0;
      while (j < l) {
        if (sel.is_member(idsi[j])) {
          l--;
          idsi[j] = idsi[l];
          memmove(codesi + j * code_size, codesi + l * code_size, code_size);
        } else {
          j++;
        }
      }
      if (l < (long)idsi.size()) {
        nremove += (long)idsi.size() - l;
        idsi.resize(l);
        codes[i].resize(l * code_size);
      }
    }
    ntotal -= nremove;
    if (nremove) ldirty_ = true;
    return nremove;
  }

  /// groups of stored vectors with identical codes in the same list (IndexIVFPQ.cpp:1239-1280): lims[0..ngroup],
  /// dup_ids[lims[g] .. lims[g+1]) = the ids of group g; returns the number of groups
  size_t find_duplicates(idx_t* dup_ids, size_t* lims) const {
    size_t ngroup = 0;
    lims[0] = 0;
    for (size_t list_no = 0; list_no < nlist; list_no++) {
      const size_t n = ids[list_no].size();
      const uint8_t* tab = codes[list_no].data();
      const size_t cs = code_size;
      std::vector<int> ord(n);
      for (size_t i = 0; i < n; i++) ord[i] = (int)i;
      auto cmp = [tab, cs](int a, int b) { return memcmp(tab + a * cs, tab + b * cs, cs); };
      std::sort(ord.begin(), ord.end(), [&](int a, int b) { return cmp(a, b) > 0; });   // CodeCmp: descending by bytes
      const long* list_ids = ids[list_no].data();
      int prev = -1;   // elements prev .. i-1 are equal
      for (size_t i = 0; i < n; i++) {
        if (prev >= 0 && cmp(ord[prev], ord[i]) == 0) {
          if ((size_t)prev + 1 == i) {   // a new group starts
            ngroup++;
            lims[ngroup] = lims[ngroup - 1];
            dup_ids[lims[ngroup]++] = list_ids[ord[prev]];
          }
          dup_ids[lims[ngroup]++] = list_ids[ord[i]];
        } else {
          prev = (int)i;
        }
      }
    }
    return ngroup;
  }

  /// moves the other index's codes behind this one's (IndexIVFPQ.cpp:327-335; the ids move in IndexIVF::merge_from)
  void merge_from_residuals(IndexIVF& other_in) override {
    IndexIVFPQ& other = dynamic_cast<IndexIVFPQ&>(other_in);
    for (size_t i = 0; i < nlist; i++) {
      codes[i].insert(codes[i].end(), other.codes[i].begin(), other.codes[i].end());
      other.codes[i].clear();
    }
    ldirty_ = other.ldirty_ = true;
  }

  /// copies the entries with a1 <= id < a2 (subset_type 0; what gpu/GpuAutoTune.cpp:231-283 shards an index with).
  /// IndexIVFPQ.h:154-158 also documents subset_type 1 (id % a1 == a2), but the reference's loop (IndexIVFPQ.cpp:337-361)
  /// copies nothing for it; the same happens here, so that a caller sees the reference's behaviour
  void copy_subset_to(IndexIVFPQ& other, int subset_type, long a1, long a2) const {
    FAISS_THROW_IF_NOT(nlist == other.nlist);
    FAISS_THROW_IF_NOT(!other.maintain_direct_map);
    const size_t cs = pq.code_size;
    for (size_t list_no = 0; list_no < nlist; list_no++) {
      const std::vector<long>& ids_in = ids[list_no];
      const std::vector<uint8_t>& codes_in = codes[list_no];
      for (size_t i = 0; i < ids_in.size(); i++) {
        const long id = ids_in[i];
        if (subset_type == 0 && a1 <= id && id < a2) {
          other.ids[list_no].push_back(id);
          other.codes[list_no].insert(other.codes[list_no].end(), codes_in.begin() + i * cs, codes_in.begin() + (i + 1) * cs);
          other.ntotal++;
        }
      }
    }
    other.ldirty_ = true;
  }
  void reconstruct(idx_t key, float* recons) const override {
    FAISS_THROW_IF_NOT(maintain_direct_map && key >= 0 && key < (idx_t)direct_map.size());
    const long list_no = direct_map[key] >> 32, offset = direct_map[key] & 0xffffffff;
    quantizer->reconstruct(list_no, recons);
    std::vector<float> r(d);
    pq.decode(&codes[list_no][offset * code_size], r.data());
    for (int i = 0; i < d; i++) recons[i] = by_residual ? recons[i] + r[i] : r[i];
  }

  /// handle of the device copy (GpuIndexIVFPQ::copyFrom reads the host fields instead)
  vlq_ivfpq_t device_handle() const { sync_(true); return h_; }

 private:
  void check_search_() const {
    FAISS_THROW_IF_NOT(is_trained);
    FAISS_THROW_IF_NOT_MSG(polysemous_ht == 0 && scan_table_threshold == 0,
                           "polysemous / on-the-fly scan modes are outside the built path");
  }
  void collect_stats_(size_t n) const {
    uint64_t nq = 0, ncode = 0;
    VLQ_CHECK(vlq_ivfpq_stats(h_, &nq, &ncode, 1));   // also raises on an invalid key (IndexIVFPQ.cpp:1008-1011)
    indexIVFPQ_stats.nq += n;
    indexIVFPQ_stats.ncode += ncode;
  }
  void sync_(bool with_lists) const {
    const IndexFlat* flat = dynamic_cast<const IndexFlat*>(quantizer);
    const MultiIndexQuantizer* miq = dynamic_cast<const MultiIndexQuantizer*>(quantizer);
    FAISS_THROW_IF_NOT_MSG((flat && flat->metric_type == METRIC_L2) || miq,
                           "coarse quantizer must be an IndexFlatL2 or a MultiIndexQuantizer");
    if (!h_) {
      VLQ_CHECK(vlq_ivfpq_create(&h_, device, d, (int)nlist, (int)pq.M, (int)pq.nbits));
      hdirty_ = ldirty_ = true;
    }
    if (hdirty_) {
      FAISS_THROW_IF_NOT(quantizer->ntotal == (idx_t)nlist);
      if (miq) VLQ_CHECK(vlq_ivfpq_set_imi_centroids(h_, (int)miq->pq.nbits, miq->pq.centroids.data()));
      else VLQ_CHECK(vlq_ivfpq_set_coarse_centroids(h_, flat->xb.data()));
      VLQ_CHECK(vlq_ivfpq_set_pq_centroids(h_, pq.centroids.data()));
      hdirty_ = false;
    }
    VLQ_CHECK(vlq_ivfpq_set_search_options(h_, by_residual, by_residual ? (use_precomputed_table ? 1 : 0) : 0,
                                           (int64_t)max_codes));
    if (with_lists && ldirty_) {
      std::vector<int64_t> off(nlist + 1, 0);
      for (size_t i = 0; i < nlist; i++) off[i + 1] = off[i] + (int64_t)ids[i].size();
      std::vector<uint8_t> fc((size_t)off[nlist] * code_size);
      std::vector<int64_t> fi((size_t)off[nlist]);
      for (size_t i = 0; i < nlist; i++) {
        if (ids[i].empty()) continue;
        memcpy(&fc[(size_t)off[i] * code_size], codes[i].data(), codes[i].size());
        for (size_t j = 0; j < ids[i].size(); j++) fi[off[i] + j] = ids[i][j];
      }
      VLQ_CHECK(vlq_ivfpq_set_lists(h_, fc.data(), fi.data(), off.data()));
      ldirty_ = false;
    }
  }
  mutable vlq_ivfpq_t h_ = nullptr;
  mutable bool hdirty_ = true, ldirty_ = true;
};

}  // namespace faiss
