// faiss::FaissException and the FAISS_THROW_* / FAISS_ASSERT* conventions of the
// reference (FaissException.h:19-36, FaissAssert.h:23-94): user errors throw,
// broken invariants print and abort.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>

namespace faiss {

class FaissException : public std::exception {
 public:
  explicit FaissException(const std::string& m) : msg(m) {}
  FaissException(const std::string& m, const char* func, const char* file, int line) {
    msg = std::string("Error in ") + func + " at " + file + ":" + std::to_string(line) + ": " + m;
  }
  const char* what() const noexcept override { return msg.c_str(); }
  std::string msg;
};

}  // namespace faiss

#define FAISS_THROW_MSG(MSG) \
  do { throw faiss::FaissException(MSG, __PRETTY_FUNCTION__, __FILE__, __LINE__); } while (false)
#define FAISS_THROW_FMT(FMT, ...)                                          \
  do {                                                                     \
    char buf_[1024];                                                       \
    snprintf(buf_, sizeof(buf_), FMT, __VA_ARGS__);                        \
    throw faiss::FaissException(buf_, __PRETTY_FUNCTION__, __FILE__, __LINE__); \
  } while (false)
#define FAISS_THROW_IF_NOT(X) \
  do { if (!(X)) { FAISS_THROW_FMT("Error: '%s' failed", #X); } } while (false)
#define FAISS_THROW_IF_NOT_MSG(X, MSG) \
  do { if (!(X)) { FAISS_THROW_FMT("Error: '%s' failed: " MSG, #X); } } while (false)
#define FAISS_THROW_IF_NOT_FMT(X, FMT, ...) \
  do { if (!(X)) { FAISS_THROW_FMT("Error: '%s' failed: " FMT, #X, __VA_ARGS__); } } while (false)
#define FAISS_ASSERT(X)                                                              \
  do {                                                                               \
    if (!(X)) {                                                                      \
      fprintf(stderr, "Faiss assertion '%s' failed in %s at %s:%d\n", #X,            \
              __PRETTY_FUNCTION__, __FILE__, __LINE__);                              \
      abort();                                                                       \
    }                                                                                \
  } while (false)

// C-ABI status -> exception (HIP failures are CUDA_VERIFY-style invariants in the
// reference, gpu/utils/DeviceUtils.h:108-116; here they surface as exceptions so a
// missing GPU is reported, not aborted on)
#include "../vlq_ivfpq.h"
#define VLQ_CHECK(EXPR)                                                                 \
  do {                                                                                  \
    int rc_ = (EXPR);                                                                   \
    if (rc_ != VLQ_OK) { FAISS_THROW_FMT("%s -> %d: %s", #EXPR, rc_, vlq_last_error()); } \
  } while (false)
