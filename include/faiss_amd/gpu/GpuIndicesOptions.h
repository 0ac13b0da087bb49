// gpu/GpuIndicesOptions.h:15-31
#pragma once
namespace faiss { namespace gpu {
enum IndicesOptions {
  INDICES_CPU = 0,     ///< reference: ids kept on the host; here ids always live in HBM, result identical
  INDICES_IVF = 1,     ///< return (list << 32 | offset) instead of the user id
  INDICES_32_BIT = 2,  ///< stored as 64 bit here (288 GB of HBM3E make the saving pointless)
  INDICES_64_BIT = 3,
};
} }
