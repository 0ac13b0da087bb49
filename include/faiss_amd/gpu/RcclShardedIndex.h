// The north star's multi-GPU mode with a C++ host: the index replicated on the G GPUs of one node, a query batch
// cut into contiguous slices of ceil(n / G) queries -- the arithmetic of faiss::gpu::IndexProxy::search
// (gpu/IndexProxy.cpp:138-149) -- every GPU answering its slice completely, and ONE RCCL all-gather per output array
// over xGMI that leaves the full (distances, labels) on every GPU.  No reduction and no merge:
// slices are disjoint, so the rows are those of one index searching the whole batch, bit for bit.
//
// This is the in-process form (one host thread, G devices, ncclCommInitAll); the one-process-per-GPU form of the same
// exchange is vector_line_quantization_amd/sharded.py + bench.py (torch.distributed, backend "nccl" = RCCL), and the
// collective-free form for callers with host buffers is gpu/IndexProxy.h.  The reference itself gathers with MPI
// point-to-point messages to rank 0 (gpu/test/deep1b16_query.cpp:354-378); RCCL's all-gather is what that becomes on
// GPUs that are linked to each other.
//
// Work per search on GPU g, all on the replica's own stream: H2D of its slice (or nothing for a device-resident
// batch on that GPU), GpuIndexIVFPQ::search with device buffers (asynchronous), ncclAllGather of the [per][k] float
// and int64 slots inside one group; then the rows are read from GPU 0.  Link with -lrccl.
#pragma once
#include <rccl/rccl.h>

#include <algorithm>
#include <limits>
#include <vector>

#include "GpuIndexIVFPQ.h"

namespace faiss { namespace gpu {

class RcclShardedIndex {
 public:
  /// slice [lo, hi) of replica g out of G for n queries, and the slot size per (IndexProxy.cpp:139-149)
  static void sliceOf(long n, int G, int g, long* lo, long* hi, long* per) {
    *per = n > 0 ? (n + G - 1) / G : 0;
    *lo = std::min<long>(n, (long)g * *per);
    *hi = std::min<long>(n, *lo + *per);
  }

  /// replicas: one trained, populated GpuIndexIVFPQ per device (distinct devices; borrowed)
  explicit RcclShardedIndex(const std::vector<GpuIndexIVFPQ*>& replicas) : rep_(replicas), st_(replicas.size()) {
    FAISS_THROW_IF_NOT_MSG(!rep_.empty(), "no replicas");
    std::vector<int> devs;
    for (auto* r : rep_) {
      FAISS_THROW_IF_NOT_MSG(r && r->d == rep_[0]->d && r->ntotal == rep_[0]->ntotal, "replicas must hold the same index");
      FAISS_THROW_IF_NOT_MSG(std::find(devs.begin(), devs.end(), r->getDevice()) == devs.end(),
                             "one replica per device (RCCL ranks of one communicator must sit on distinct GPUs)");
      devs.push_back(r->getDevice());
    }
    comms_.resize(rep_.size());
    if (ncclCommInitAll(comms_.data(), (int)rep_.size(), devs.data()) != ncclSuccess) {
      comms_.clear();
      FAISS_THROW_MSG("ncclCommInitAll failed");
    }
  }
  ~RcclShardedIndex() {
    for (size_t g = 0; g < st_.size(); g++) {
      (void)hipSetDevice(rep_[g]->getDevice());
      for (void* p : {st_[g].x, st_[g].Ds, st_[g].Is, st_[g].Dall, st_[g].Iall})
        if (p) (void)hipFree(p);
    }
    for (auto c : comms_) (void)ncclCommDestroy(c);
  }
  RcclShardedIndex(const RcclShardedIndex&) = delete;
  RcclShardedIndex& operator=(const RcclShardedIndex&) = delete;

  int numReplicas() const { return (int)rep_.size(); }

  /// x [n][d] host memory; distances [n][k], labels [n][k] host memory.  nprobe is each replica's own setting.
  void search(Index::idx_t n, const float* x, Index::idx_t k, float* distances, Index::idx_t* labels) {
    FAISS_THROW_IF_NOT_MSG(k >= 1 && k <= VLQ_MAX_K, "k outside 1..1024");
    if (n == 0) return;
    const int G = (int)rep_.size(), d = rep_[0]->d;
    long per = 0;
    for (int g = 0; g < G; g++) {
      long lo, hi;
      sliceOf(n, G, g, &lo, &hi, &per);
      Buf& b = st_[g];
      check(hipSetDevice(rep_[g]->getDevice()));
      hipStream_t s = rep_[g]->getResources()->getDefaultStream(rep_[g]->getDevice());
      reserve(b, (size_t)per, (size_t)G, (size_t)d, (size_t)k);
      // empty slots keep the padding of an index with fewer than k results (Heap.h:318-321)
      if (hi - lo < per) {
        fill(b.Ds, std::numeric_limits<float>::max(), (size_t)per * k, s);
        check(hipMemsetAsync(b.Is, 0xff, (size_t)per * k * sizeof(Index::idx_t), s));
      }
      if (hi > lo) {
        check(hipMemcpyAsync(b.x, x + (size_t)lo * d, (size_t)(hi - lo) * d * sizeof(float), hipMemcpyHostToDevice, s));
        rep_[g]->search(hi - lo, (const float*)b.x, k, (float*)b.Ds, (Index::idx_t*)b.Is);   // device buffers: asynchronous
      }
    }
    // one all-gather per output array, all ranks in one group
    ncclCheck(ncclGroupStart());
    for (int g = 0; g < G; g++) {
      hipStream_t s = rep_[g]->getResources()->getDefaultStream(rep_[g]->getDevice());
      ncclCheck(ncclAllGather(st_[g].Ds, st_[g].Dall, (size_t)per * k, ncclFloat, comms_[g], s));
      ncclCheck(ncclAllGather(st_[g].Is, st_[g].Iall, (size_t)per * k, ncclInt64, comms_[g], s));
    }
    ncclCheck(ncclGroupEnd());
    // every GPU now holds all rows: read them from the first
    check(hipSetDevice(rep_[0]->getDevice()));
    hipStream_t s0 = rep_[0]->getResources()->getDefaultStream(rep_[0]->getDevice());
    check(hipMemcpyAsync(distances, st_[0].Dall, (size_t)n * k * sizeof(float), hipMemcpyDeviceToHost, s0));
    check(hipMemcpyAsync(labels, st_[0].Iall, (size_t)n * k * sizeof(Index::idx_t), hipMemcpyDeviceToHost, s0));
    for (int g = 0; g < G; g++) {
      check(hipSetDevice(rep_[g]->getDevice()));
      check(hipStreamSynchronize(rep_[g]->getResources()->getDefaultStream(rep_[g]->getDevice())));
    }
  }

  /// device pointer of the gathered rows on replica g after search(): [G * per][k] (tests: every GPU holds them)
  const float* gatheredDistances(int g) const { return (const float*)st_[g].Dall; }
  const Index::idx_t* gatheredLabels(int g) const { return (const Index::idx_t*)st_[g].Iall; }

 private:
  struct Buf {
    void *x = nullptr, *Ds = nullptr, *Is = nullptr, *Dall = nullptr, *Iall = nullptr;
    size_t cx = 0, cDs = 0, cIs = 0, cDall = 0, cIall = 0;      // capacities in bytes
  };
  static void check(hipError_t e) {
    if (e != hipSuccess) FAISS_THROW_FMT("HIP error: %s", hipGetErrorString(e));
  }
  static void ncclCheck(ncclResult_t r) {
    if (r != ncclSuccess) FAISS_THROW_FMT("RCCL error: %s", ncclGetErrorString(r));
  }
  static void grow(void*& p, size_t& cap, size_t bytes) {
    if (bytes <= cap) return;
    if (p) check(hipFree(p));
    p = nullptr;
    check(hipMalloc(&p, bytes));
    cap = bytes;
  }
  static void reserve(Buf& b, size_t per, size_t G, size_t d, size_t k) {
    grow(b.x, b.cx, per * d * sizeof(float));
    grow(b.Ds, b.cDs, per * k * sizeof(float));
    grow(b.Is, b.cIs, per * k * sizeof(Index::idx_t));
    grow(b.Dall, b.cDall, G * per * k * sizeof(float));
    grow(b.Iall, b.cIall, G * per * k * sizeof(Index::idx_t));
  }
  static void fill(void* p, float v, size_t n, hipStream_t s) {
    std::vector<float> h(n, v);           // padding of a short last slice: rare, small
    check(hipMemcpyAsync(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice, s));
    check(hipStreamSynchronize(s));
  }

  std::vector<GpuIndexIVFPQ*> rep_;
  std::vector<Buf> st_;
  std::vector<ncclComm_t> comms_;
};

} }
