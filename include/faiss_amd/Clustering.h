// k-means with the reference's recipe (Clustering.cpp:27-35, :66-206, utils.cpp
// km_update_centroids :1369-1450): subsample to max_points_per_centroid, initialise
// from a seeded random permutation, Lloyd iterations whose ASSIGNMENT step is the
// index's search() -- here the MFMA coarse kernel through the C ABI -- centroid
// update on the host, empty clusters re-seeded by splitting a populated one with a
// +-1/1024 perturbation.  Training is outside the hot path; it
// exists so the Index API is usable end to end.  The random stream is our own
// (splitmix64), so centroids differ from the reference's for the same seed.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "Index.h"

namespace faiss {

struct ClusteringParameters {
  int niter;
  int nredo;
  bool verbose;
  bool spherical;
  bool update_index;
  int min_points_per_centroid;
  int max_points_per_centroid;
  int seed;
  ClusteringParameters()
      : niter(25), nredo(1), verbose(false), spherical(false), update_index(false),
        min_points_per_centroid(39), max_points_per_centroid(256), seed(1234) {}
};

namespace detail {
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  size_t rand_int(size_t n) { return (size_t)(next() % n); }
  float rand_float() { return (next() >> 40) * (1.0f / 16777216.0f); }
};
inline void rand_perm(std::vector<int>& perm, size_t n, uint64_t seed) {
  perm.resize(n);
  for (size_t i = 0; i < n; i++) perm[i] = (int)i;
  SplitMix64 rng(seed);
  for (size_t i = 0; i + 1 < n; i++) std::swap(perm[i], perm[i + rng.rand_int(n - i)]);
}
}  // namespace detail

/// subsample to at most nmax vectors (fvecs_maybe_subsample, utils.cpp:1941-1962)
inline std::vector<float> maybe_subsample(size_t d, size_t* n, size_t nmax, const float* x, uint64_t seed) {
  std::vector<float> out;
  if (*n <= nmax) { out.assign(x, x + *n * d); return out; }
  std::vector<int> perm;
  detail::rand_perm(perm, *n, seed);
  out.resize(nmax * d);
  for (size_t i = 0; i < nmax; i++) memcpy(&out[i * d], x + (size_t)perm[i] * d, sizeof(float) * d);
  *n = nmax;
  return out;
}

struct Clustering : ClusteringParameters {
  typedef Index::idx_t idx_t;
  size_t d, k;
  std::vector<float> centroids;
  std::vector<float> obj;

  Clustering(int d, int k) : d(d), k(k) {}
  Clustering(int d, int k, const ClusteringParameters& cp) : ClusteringParameters(cp), d(d), k(k) {}
  virtual ~Clustering() {}

  static int update_centroids(const float* x, float* cent, const idx_t* assign, size_t d, size_t k, size_t n) {
    std::vector<size_t> h(k, 0);
    std::vector<double> acc(k * d, 0.0);
    for (size_t i = 0; i < n; i++) {
      const idx_t c = assign[i];
      h[c]++;
      double* a = &acc[(size_t)c * d];
      const float* xi = x + i * d;
      for (size_t j = 0; j < d; j++) a[j] += xi[j];
    }
    for (size_t c = 0; c < k; c++)
      if (h[c]) for (size_t j = 0; j < d; j++) cent[c * d + j] = (float)(acc[c * d + j] / (double)h[c]);
    int nsplit = 0;
    detail::SplitMix64 rng(1234);
    const float eps = 1.f / 1024.f;
    for (size_t ci = 0; ci < k; ci++) {
      if (h[ci] != 0) continue;
      size_t cj = 0;
      for (;; cj = (cj + 1) % k) {
        const float p = ((float)h[cj] - 1.0f) / (float)(n - k);
        if (rng.rand_float() < p) break;
      }
      for (size_t j = 0; j < d; j++) {
        const float v = cent[cj * d + j];
        cent[ci * d + j] = v * ((j % 2 == 0) ? 1 + eps : 1 - eps);
        cent[cj * d + j] = v * ((j % 2 == 0) ? 1 - eps : 1 + eps);
      }
      h[ci] = h[cj] / 2;
      h[cj] -= h[ci];
      nsplit++;
    }
    return nsplit;
  }

  /// On return the centroids are also added to `index` (Clustering.cpp:66-206).
  virtual void train(idx_t nx, const float* x_in, Index& index) {
    FAISS_THROW_IF_NOT_MSG(nx >= (idx_t)k, "need at least as many training points as clusters");
    for (size_t i = 0; i < (size_t)nx * d; i++)
      FAISS_THROW_IF_NOT_MSG(std::isfinite(x_in[i]), "input contains NaN's or Inf's");
    size_t n = nx;
    std::vector<float> sub;
    const float* x = x_in;
    if (n > k * (size_t)max_points_per_centroid) {
      sub = maybe_subsample(d, &n, k * (size_t)max_points_per_centroid, x_in, seed);
      x = sub.data();
    } else if (n < k * (size_t)min_points_per_centroid) {
      fprintf(stderr, "WARNING clustering %zu points to %zu centroids: please provide at least %zu training points\n",
              n, k, k * (size_t)min_points_per_centroid);
    }
    std::vector<idx_t> assign(n);
    std::vector<float> dis(n);
    if (centroids.empty()) {
      centroids.resize(d * k);
      std::vector<int> perm;
      detail::rand_perm(perm, n, (uint64_t)seed + 1);
      for (size_t i = 0; i < k; i++) memcpy(&centroids[i * d], x + (size_t)perm[i] * d, d * sizeof(float));
    } else {
      FAISS_THROW_IF_NOT(centroids.size() == d * k);
    }
    index.reset();
    index.train(k, centroids.data());
    index.add(k, centroids.data());
    for (int it = 0; it < niter; it++) {
      index.search(n, x, 1, dis.data(), assign.data());
      double err = 0;
      for (size_t j = 0; j < n; j++) err += dis[j];
      obj.push_back((float)err);
      const int nsplit = update_centroids(x, centroids.data(), assign.data(), d, k, n);
      if (verbose) printf("  Iteration %d: objective=%g nsplit=%d\n", it, err, nsplit);
      index.reset();
      index.add(k, centroids.data());
    }
  }
};

}  // namespace faiss
