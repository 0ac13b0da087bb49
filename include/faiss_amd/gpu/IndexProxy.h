// faiss::gpu::IndexProxy (gpu/IndexProxy.h:22-84, gpu/IndexProxy.cpp:20-168): the reference's
// replica mode for one process driving several GPUs -- the same index on every device, a query
// batch cut into ceil(n / #indices) contiguous slices, every replica answering its slice straight
// into the caller's distances / labels.  No collective and no merge: slices are disjoint, so the
// result is what one index returns for the whole batch.
//
// Same public surface (addIndex / removeIndex / runOnIndex / reset / train / add / search /
// reconstruct / own_fields / count / at).  Each replica is served by its own host thread for the
// proxy's lifetime, as in the reference (utils/WorkerThread.h) -- here a small task loop on
// std::thread; a replica's calls therefore always come from the same thread, which is what a
// device-bound index (hipSetDevice per call, one stream) wants.
//
// The multi-PROCESS form of the same slicing (one rank per GPU, RCCL all-gather of the slices) is
// vector_line_quantization_amd/sharded.py + bench.py; this header is the multi-GPU path of a C++
// driver that links the library directly.
#pragma once
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "compat.h"

namespace faiss { namespace gpu {

class IndexProxy : public faiss::Index {
 public:
  IndexProxy() : own_fields(false) {}
  ~IndexProxy() override {
    lanes_.clear();                       // joins every replica's thread first
    if (own_fields)
      for (auto* ix : indices_) delete ix;
  }

  /// Adds a replica.  From here on it must only be touched through the proxy (runOnIndex).
  void addIndex(faiss::Index* index) {
    FAISS_THROW_IF_NOT_MSG(index, "null index");
    if (!indices_.empty()) {
      const faiss::Index* first = indices_.front();
      // the reference asserts; a mismatch is a caller error, so it is reported as one
      FAISS_THROW_IF_NOT_MSG(index->d == first->d, "replicas must have the same dimension");
      FAISS_THROW_IF_NOT_MSG(index->ntotal == first->ntotal, "replicas must hold the same number of vectors");
      FAISS_THROW_IF_NOT_MSG(index->metric_type == first->metric_type, "replicas must use the same metric");
    } else {
      d = index->d;
      ntotal = index->ntotal;
      verbose = index->verbose;
      is_trained = index->is_trained;
      metric_type = index->metric_type;
    }
    indices_.push_back(index);
    lanes_.emplace_back(new Lane());
  }

  /// Flushes the replica's pending work, stops its thread and forgets it.
  void removeIndex(faiss::Index* index) {
    for (size_t i = 0; i < indices_.size(); i++)
      if (indices_[i] == index) {
        lanes_.erase(lanes_.begin() + (long)i);      // ~Lane drains and joins
        indices_.erase(indices_.begin() + (long)i);
        return;
      }
    FAISS_THROW_MSG("removeIndex: index is not managed by this proxy");
  }

  /// f(index) on every replica, each in its own thread; returns when all are done.  The first
  /// exception thrown by a replica is rethrown here.
  void runOnIndex(std::function<void(faiss::Index*)> f) {
    std::vector<std::future<void> > done;
    for (size_t i = 0; i < indices_.size(); i++) {
      faiss::Index* ix = indices_[i];
      done.push_back(lanes_[i]->run([f, ix]() { f(ix); }));
    }
    wait_all(done);
  }

  void reset() override {
    runOnIndex([](faiss::Index* ix) { ix->reset(); });
    ntotal = 0;
  }
  void train(idx_t n, const float* x) override {
    runOnIndex([n, x](faiss::Index* ix) { ix->train(n, x); });
    if (!indices_.empty()) is_trained = indices_.front()->is_trained;
  }
  void add(idx_t n, const float* x) override {
    runOnIndex([n, x](faiss::Index* ix) { ix->add(n, x); });
    ntotal += n;
  }

  /// slice i = queries [i * per, min(n, (i + 1) * per)), per = ceil(n / #replicas)
  /// (gpu/IndexProxy.cpp:139-149); replicas beyond the last non-empty slice stay idle
  void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override {
    FAISS_THROW_IF_NOT_MSG(!indices_.empty(), "IndexProxy::search without replicas");
    if (n == 0) return;
    const idx_t nrep = (idx_t)indices_.size();
    const idx_t per = (n + nrep - 1) / nrep;
    const int dim = indices_.front()->d;
    std::vector<std::future<void> > done;
    for (idx_t i = 0; i < nrep; i++) {
      const idx_t base = i * per;
      if (base >= n) break;
      const idx_t cnt = per < n - base ? per : n - base;
      faiss::Index* ix = indices_[(size_t)i];
      const float* xs = x + base * dim;
      float* ds = distances + base * k;
      idx_t* ls = labels + base * k;
      done.push_back(lanes_[(size_t)i]->run([ix, cnt, xs, k, ds, ls]() { ix->search(cnt, xs, k, ds, ls); }));
    }
    wait_all(done);
  }

  /// bounds of slice `i` of an n-query batch over `nrep` replicas (what search() uses)
  static void sliceOf(idx_t n, idx_t nrep, idx_t i, idx_t* base, idx_t* count) {
    const idx_t per = nrep > 0 ? (n + nrep - 1) / nrep : 0;
    const idx_t b = i * per < n ? i * per : n;
    *base = b;
    *count = per < n - b ? per : n - b;
  }

  /// from the first replica
  void reconstruct(idx_t key, float* v) const override {
    FAISS_THROW_IF_NOT_MSG(!indices_.empty(), "IndexProxy::reconstruct without replicas");
    indices_.front()->reconstruct(key, v);
  }

  bool own_fields;
  int count() const { return (int)indices_.size(); }
  faiss::Index* at(int i) { return indices_[(size_t)i]; }
  const faiss::Index* at(int i) const { return indices_[(size_t)i]; }

 private:
  // one host thread with a FIFO of tasks
  class Lane {
   public:
    Lane() : stop_(false), th_([this]() { loop(); }) {}
    ~Lane() {
      {
        std::lock_guard<std::mutex> g(mu_);
        stop_ = true;
      }
      cv_.notify_all();
      th_.join();                         // pending tasks are run before the thread leaves
    }
    std::future<void> run(std::function<void()> fn) {
      std::packaged_task<void()> task(std::move(fn));
      std::future<void> fut = task.get_future();
      {
        std::lock_guard<std::mutex> g(mu_);
        q_.push_back(std::move(task));
      }
      cv_.notify_one();
      return fut;
    }

   private:
    void loop() {
      for (;;) {
        std::packaged_task<void()> task;
        {
          std::unique_lock<std::mutex> g(mu_);
          cv_.wait(g, [this]() { return stop_ || !q_.empty(); });
          if (q_.empty()) return;         // stop requested and nothing left
          task = std::move(q_.front());
          q_.pop_front();
        }
        task();                           // exceptions land in the future
      }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::packaged_task<void()> > q_;
    bool stop_;
    std::thread th_;
  };

  static void wait_all(std::vector<std::future<void> >& done) {
    std::exception_ptr first;
    for (auto& f : done) {
      try { f.get(); } catch (...) { if (!first) first = std::current_exception(); }
    }
    if (first) std::rethrow_exception(first);
  }

  std::vector<faiss::Index*> indices_;
  mutable std::vector<std::unique_ptr<Lane> > lanes_;
};

} }
