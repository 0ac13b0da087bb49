// write_index / read_index for the index types on the hot path, in the reference's
// on-disk format (index_io.cpp:226-317, :459-532, :536-680): fourcc + header, nested
// quantizer, per-list id and code vectors.
//   "IxF2"  IndexFlatL2            "Imiq"  MultiIndexQuantizer        "IvPQ"  IndexIVFPQ
// Files written by the reference load here and vice versa (tests/test_index_io.py checks
// byte identity against files the reference wrote).  Other fourccs are outside the path
// and rejected.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <vector>

#include "IndexFlat.h"
#include "IndexIVFPQ.h"
#include "IndexPQ.h"

namespace faiss {

namespace io_detail {

inline uint32_t fourcc(const char sx[4]) {
  const unsigned char* x = (const unsigned char*)sx;
  return x[0] | x[1] << 8 | x[2] << 16 | (uint32_t)x[3] << 24;
}
struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };

template <typename T> void w1(FILE* f, const T& x) {
  FAISS_THROW_IF_NOT_MSG(fwrite(&x, sizeof(T), 1, f) == 1, "write error");
}
template <typename T> void r1(FILE* f, T& x) {
  FAISS_THROW_IF_NOT_MSG(fread(&x, sizeof(T), 1, f) == 1, "read error");
}
template <typename T> void wvec(FILE* f, const std::vector<T>& v) {
  size_t n = v.size();
  w1(f, n);
  FAISS_THROW_IF_NOT_MSG(fwrite(v.data(), sizeof(T), n, f) == n, "write error");
}
template <typename T> void rvec(FILE* f, std::vector<T>& v) {
  long n;
  r1(f, n);
  FAISS_THROW_IF_NOT(n >= 0 && n < (1L << 40));
  v.resize(n);
  FAISS_THROW_IF_NOT_MSG(fread(v.data(), sizeof(T), n, f) == (size_t)n, "read error");
}
inline void write_header(const Index* idx, FILE* f) {   // index_io.cpp:147-155
  w1(f, idx->d);
  w1(f, idx->ntotal);
  Index::idx_t dummy = 1 << 20;
  w1(f, dummy);
  w1(f, dummy);
  w1(f, idx->is_trained);
  w1(f, idx->metric_type);
}
inline void read_header(Index* idx, FILE* f) {
  r1(f, idx->d);
  r1(f, idx->ntotal);
  Index::idx_t dummy;
  r1(f, dummy);
  r1(f, dummy);
  r1(f, idx->is_trained);
  r1(f, idx->metric_type);
}
inline void write_pq(const ProductQuantizer& pq, FILE* f) {
  w1(f, pq.d); w1(f, pq.M); w1(f, pq.nbits);
  wvec(f, pq.centroids);
}
inline void read_pq(ProductQuantizer& pq, FILE* f) {
  r1(f, pq.d); r1(f, pq.M); r1(f, pq.nbits);
  pq.set_derived_values();
  rvec(f, pq.centroids);
}

}  // namespace io_detail

inline void write_index(const Index* idx, FILE* f) {
  using namespace io_detail;
  if (const IndexFlat* flat = dynamic_cast<const IndexFlat*>(idx)) {
    FAISS_THROW_IF_NOT_MSG(flat->metric_type == METRIC_L2, "only IndexFlatL2 is on the path");
    w1(f, fourcc("IxF2"));
    write_header(idx, f);
    wvec(f, flat->xb);
  } else if (const MultiIndexQuantizer* miq = dynamic_cast<const MultiIndexQuantizer*>(idx)) {
    w1(f, fourcc("Imiq"));
    write_header(idx, f);
    write_pq(miq->pq, f);
  } else if (const IndexIVFPQ* ivpq = dynamic_cast<const IndexIVFPQ*>(idx)) {
    w1(f, fourcc("IvPQ"));
    write_header(idx, f);                      // write_ivf_header, index_io.cpp:226-238
    w1(f, ivpq->nlist);
    w1(f, ivpq->nprobe);
    write_index(ivpq->quantizer, f);
    for (size_t i = 0; i < ivpq->nlist; i++) wvec(f, ivpq->ids[i]);
    w1(f, ivpq->maintain_direct_map);
    wvec(f, ivpq->direct_map);
    w1(f, ivpq->by_residual);
    w1(f, ivpq->code_size);
    write_pq(ivpq->pq, f);
    for (size_t i = 0; i < ivpq->codes.size(); i++) wvec(f, ivpq->codes[i]);
  } else {
    FAISS_THROW_MSG("don't know how to serialize this type of index");
  }
}

inline void write_index(const Index* idx, const char* fname) {
  FILE* f = fopen(fname, "w");
  FAISS_THROW_IF_NOT_MSG(f, "cannot open file for writing");
  io_detail::FileCloser c{f};
  write_index(idx, f);
}

/// precompute = false skips IndexIVFPQ::precompute_table (which runs on the device)
inline Index* read_index(FILE* f, bool precompute = true) {
  using namespace io_detail;
  uint32_t h;
  r1(f, h);
  if (h == fourcc("IxF2")) {
    std::unique_ptr<IndexFlatL2> flat(new IndexFlatL2());
    read_header(flat.get(), f);
    rvec(f, flat->xb);
    FAISS_THROW_IF_NOT(flat->xb.size() == (size_t)flat->ntotal * flat->d);
    return flat.release();
  }
  if (h == fourcc("Imiq")) {
    std::unique_ptr<MultiIndexQuantizer> miq(new MultiIndexQuantizer(2, 2, 1));
    read_header(miq.get(), f);
    read_pq(miq->pq, f);
    return miq.release();
  }
  if (h == fourcc("IvPQ")) {
    // a minimal quantizer to construct with; replaced by the stored one
    IndexFlatL2* tmpq = new IndexFlatL2(1);
    std::unique_ptr<IndexIVFPQ> iv(new IndexIVFPQ(tmpq, 1, 1, 1, 1));
    delete tmpq;
    read_header(iv.get(), f);
    r1(f, iv->nlist);
    r1(f, iv->nprobe);
    iv->quantizer = read_index(f, precompute);
    iv->own_fields = true;
    iv->ids.resize(iv->nlist);
    for (size_t i = 0; i < iv->nlist; i++) rvec(f, iv->ids[i]);
    r1(f, iv->maintain_direct_map);
    rvec(f, iv->direct_map);
    r1(f, iv->by_residual);
    r1(f, iv->code_size);
    read_pq(iv->pq, f);
    iv->codes.resize(iv->nlist);
    for (size_t i = 0; i < iv->nlist; i++) rvec(f, iv->codes[i]);
    iv->quantizer_trains_alone = dynamic_cast<MultiIndexQuantizer*>(iv->quantizer) != nullptr;
    // the precomputed table is not stored, it is recomputed (index_io.cpp:491-495)
    iv->use_precomputed_table = 0;
    if (iv->by_residual && precompute) iv->precompute_table();
    return iv.release();
  }
  FAISS_THROW_MSG("index type (fourcc) outside the IVFPQ path: not built");
}

inline Index* read_index(const char* fname, bool precompute = true) {
  FILE* f = fopen(fname, "r");
  FAISS_THROW_IF_NOT_MSG(f, "cannot open file for reading");
  io_detail::FileCloser c{f};
  return read_index(f, precompute);
}

}  // namespace faiss
