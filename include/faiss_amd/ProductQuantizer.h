// faiss::ProductQuantizer data model (ProductQuantizer.h:25-60) with training by
// per-sub-quantizer k-means (ProductQuantizer.cpp:236-308, Train_default).
#pragma once
#include <cstdint>
#include <vector>

#include "Clustering.h"
#include "IndexFlat.h"

namespace faiss {

struct ProductQuantizer {
  size_t d, M, nbits;
  size_t dsub, byte_per_idx, code_size, ksub;
  bool verbose;
  ClusteringParameters cp;
  std::vector<float> centroids;   ///< M * ksub * dsub

  ProductQuantizer(size_t d, size_t M, size_t nbits) : d(d), M(M), nbits(nbits) { set_derived_values(); }
  ProductQuantizer() : d(0), M(1), nbits(0) { set_derived_values(); }

  void set_derived_values() {   // ProductQuantizer.cpp:163-175
    FAISS_THROW_IF_NOT(M > 0 && d % M == 0);
    dsub = d / M;
    byte_per_idx = (nbits + 7) / 8;
    code_size = byte_per_idx * M;
    ksub = (size_t)1 << nbits;
    centroids.resize(d * ksub);
    verbose = false;
  }
  float* get_centroids(size_t m, size_t i) { return &centroids[(m * ksub + i) * dsub]; }
  const float* get_centroids(size_t m, size_t i) const { return &centroids[(m * ksub + i) * dsub]; }

  void train(int n, const float* x) {
    std::vector<float> xs((size_t)n * dsub);
    for (size_t m = 0; m < M; m++) {
      for (int j = 0; j < n; j++) memcpy(&xs[(size_t)j * dsub], x + (size_t)j * d + m * dsub, dsub * sizeof(float));
      Clustering clus((int)dsub, (int)ksub, cp);
      IndexFlatL2 index(dsub);
      clus.train(n, xs.data(), index);
      memcpy(get_centroids(m, 0), clus.centroids.data(), ksub * dsub * sizeof(float));
    }
  }
  /// fvec_L2sqr in the reference's operation order (utils.cpp:481-506): four lane sums s[l] += (x-y)^2 with multiply and
  /// add kept apart, zero-padded tail, then (s0+s1)+(s2+s3) -- the order the device encoder uses (sse_order.cuh), so that
  /// encode() and encode_multiple() pick the same centroid on near-ties.  No FMA contraction whatever the host flags.
#if defined(__clang__)
  static float l2sqr_lane_order(const float* x, const float* y, size_t d) {
#pragma clang fp contract(off)
#elif defined(__GNUC__)
  __attribute__((optimize("fp-contract=off"))) static float l2sqr_lane_order(const float* x, const float* y, size_t d) {
#else
  static float l2sqr_lane_order(const float* x, const float* y, size_t d) {
#endif
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    size_t i = 0;
    for (; i + 4 <= d; i += 4)
      for (size_t l = 0; l < 4; l++) { const float t = x[i + l] - y[i + l]; const float p = t * t; s[l] = s[l] + p; }
    if (i < d) {
      for (size_t l = 0; l < 4; l++) {
        const float t = (i + l < d) ? x[i + l] - y[i + l] : 0.f;
        const float p = t * t;
        s[l] = s[l] + p;
      }
    }
    const float a = s[0] + s[1], b = s[2] + s[3];
    return a + b;
  }
  /// nearest centroid per sub-quantizer, first minimum wins (ProductQuantizer.cpp:311-336: fvec_L2sqr_ny + strict <).
  /// Host-side, one vector at a time; bulk encoding goes through the device (IndexIVFPQ::encode_multiple).
  void compute_code(const float* x, uint8_t* code) const {
    FAISS_THROW_IF_NOT_MSG(byte_per_idx == 1, "only one byte per sub-quantizer index is built");
    for (size_t m = 0; m < M; m++) {
      const float* xs = x + m * dsub;
      float best = 3.402823466e+38f;
      size_t bi = 0;
      for (size_t j = 0; j < ksub; j++) {
        const float dis = l2sqr_lane_order(xs, get_centroids(m, j), dsub);
        if (dis < best) { best = dis; bi = j; }
      }
      code[m] = (uint8_t)bi;
    }
  }
  void decode(const uint8_t* code, float* x) const {
    for (size_t m = 0; m < M; m++) memcpy(x + m * dsub, get_centroids(m, code[m]), sizeof(float) * dsub);
  }
  void decode(const uint8_t* code, float* x, size_t n) const {
    for (size_t i = 0; i < n; i++) decode(code + M * i, x + d * i);
  }
};

}  // namespace faiss
