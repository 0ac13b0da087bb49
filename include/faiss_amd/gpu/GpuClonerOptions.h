// faiss::gpu::GpuClonerOptions / GpuMultipleClonerOptions (gpu/GpuClonerOptions.h:15-52, defaults
// gpu/GpuClonerOptions.cpp:14-26): the option block the reference's drivers fill in before they build a
// GpuIndexIVFPQConfig from it (gpu/test/deep1b16_query.cpp:236-243).  Same fields, same defaults.
// What the MI355X library does with them:
//   indicesOptions            ids always live in HBM as 64-bit values; INDICES_IVF is honoured (pairs)
//   useFloat16CoarseQuantizer accepted, computed in fp32 (superset precision; see GpuIndexIVFPQ.h)
//   useFloat16                16 x 8-bit codes: float16 look-up tables, built as the reference builds them
//                             (vlq_line_set_float16_tables / vlq_ivfpq_set_float16_tables); other shapes: fp32
//   usePrecomputed            honoured (table mode 1)
//   reserveVecs               honoured (reserveMemory)
//   storeTransposed           layout hint of the reference's cuBLAS call: no effect
#pragma once
#include "GpuIndicesOptions.h"

namespace faiss { namespace gpu {

struct GpuClonerOptions {
  GpuClonerOptions()
      : indicesOptions(INDICES_64_BIT), useFloat16CoarseQuantizer(false), useFloat16(false),
        usePrecomputed(true), reserveVecs(0), storeTransposed(false), verbose(false) {}
  IndicesOptions indicesOptions;
  bool useFloat16CoarseQuantizer;
  bool useFloat16;
  bool usePrecomputed;
  long reserveVecs;
  bool storeTransposed;
  bool verbose;
};

struct GpuMultipleClonerOptions : public GpuClonerOptions {
  GpuMultipleClonerOptions() : shard(false) {}
  bool shard;     ///< shard the lists over the GPUs instead of replicating (IndexShards vs IndexProxy)
};

} }
