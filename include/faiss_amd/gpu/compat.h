// Which faiss CPU headers the GPU shell is compiled against:
//   default                   : this repository's include/faiss_amd/*.h
//   -DVLQ_WITH_REFERENCE_FAISS: the reference's own Index.h / IndexFlat.h / IndexIVFPQ.h /
//                               FaissAssert.h (put the reference tree on the include
//                               path) -- the integration of INTEGRATION.md §A.
#pragma once
#ifdef VLQ_WITH_REFERENCE_FAISS
#include <FaissAssert.h>
#include <IndexFlat.h>
#include <IndexIVFPQ.h>
#include "../../vlq_ivfpq.h"
#ifndef VLQ_CHECK
#define VLQ_CHECK(EXPR)                                                                 \
  do {                                                                                  \
    int rc_ = (EXPR);                                                                   \
    if (rc_ != VLQ_OK) { FAISS_THROW_FMT("%s -> %d: %s", #EXPR, rc_, vlq_last_error()); } \
  } while (false)
#endif
#else
#include "../FaissException.h"
#include "../IndexFlat.h"
#include "../IndexIVFPQ.h"
#endif
