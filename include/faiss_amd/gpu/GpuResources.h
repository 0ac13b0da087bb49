// faiss::gpu::GpuResources / StandardGpuResources (gpu/GpuResources.h:23-55,
// gpu/StandardGpuResources.h:23-84).  The reference hands out cuBLAS handles, a
// scratch stack (30 % of device memory) and pinned memory; the MI355X library owns
// its workspace (sized on demand out of 288 GB of HBM3E) and needs only streams, so
// the temp-memory setters are accepted and recorded but allocate nothing.  Pinned memory
// is real: getPinnedMemory() hands out one page-locked block (256 MB unless setPinnedMemory
// said otherwise, StandardGpuResources.cpp:24), allocated on first use.  A caller that keeps
// its queries and result arrays there gets truly asynchronous copies: the library's kernels
// are enqueued behind the copy instead of after it (a pageable hipMemcpyAsync blocks the host).
#pragma once
#include <cstddef>
#include <map>
#include <utility>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "compat.h"

namespace faiss { namespace gpu {

class GpuResources {
 public:
  virtual ~GpuResources() {}
  virtual void initializeForDevice(int device) = 0;
  virtual hipStream_t getDefaultStream(int device) = 0;
  virtual std::vector<hipStream_t> getAlternateStreams(int device) = 0;
  virtual hipStream_t getAsyncCopyStream(int device) = 0;
  virtual std::pair<void*, size_t> getPinnedMemory() = 0;
};

class StandardGpuResources : public GpuResources {
 public:
  StandardGpuResources() : tempMemSize_(0), pinnedSize_((size_t)256 << 20), pinned_(nullptr) {}
  ~StandardGpuResources() override {
    for (auto& kv : streams_) for (auto s : kv.second) (void)hipStreamDestroy(s);
    if (pinned_) (void)hipHostFree(pinned_);
  }
  void noTempMemory() { tempMemSize_ = 0; }
  void setTempMemory(size_t size) { tempMemSize_ = size; }
  void setTempMemoryFraction(float) {}
  /// StandardGpuResources.cpp:60-67: only before the block exists
  void setPinnedMemory(size_t size) {
    FAISS_THROW_IF_NOT_MSG(!pinned_, "pinned memory already allocated");
    pinnedSize_ = size;
  }

  void initializeForDevice(int device) override {
    if (streams_.count(device)) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) FAISS_THROW_FMT("hipSetDevice(%d) failed", device);
    std::vector<hipStream_t> v(4);   // default, 2 alternates, async copy (StandardGpuResources.cpp:126-148)
    for (auto& s : v)
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) FAISS_THROW_MSG("hipStreamCreate failed");
    streams_[device] = v;
    (void)hipSetDevice(prev);
  }
  hipStream_t getDefaultStream(int device) override { initializeForDevice(device); return streams_[device][0]; }
  std::vector<hipStream_t> getAlternateStreams(int device) override {
    initializeForDevice(device);
    return {streams_[device][1], streams_[device][2]};
  }
  hipStream_t getAsyncCopyStream(int device) override { initializeForDevice(device); return streams_[device][3]; }
  std::pair<void*, size_t> getPinnedMemory() override {
    if (!pinned_ && pinnedSize_ > 0) {
      if (hipHostMalloc(&pinned_, pinnedSize_, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        pinned_ = nullptr;
        FAISS_THROW_FMT("hipHostMalloc(%zu) failed", pinnedSize_);
      }
    }
    return {pinned_, pinned_ ? pinnedSize_ : 0};
  }

 private:
  std::map<int, std::vector<hipStream_t> > streams_;
  size_t tempMemSize_, pinnedSize_;
  void* pinned_;
};

} }
