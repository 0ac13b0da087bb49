// Compile/link check of INTEGRATION.md §A: the GPU shell compiled against the
// REFERENCE's own CPU headers and linked with the reference's own CPU library
// (oracle/_ref/libfaiss_ref.so).  Built only where /root/reference exists; it is run
// only as far as constructing the CPU index (no GPU in the build container).
#include <cstdio>
#include <vector>

#include "faiss_amd/gpu/GpuIndexIVFPQ.h"
#include "faiss_amd/gpu/StandardGpuResources.h"

int main(int argc, char**) {
  const int d = 32, nlist = 8;
  faiss::IndexFlatL2 quantizer(d);                     // reference class
  faiss::IndexIVFPQ cpu(&quantizer, d, nlist, 8, 8);   // reference class
  std::vector<float> x(2000 * d);
  for (size_t i = 0; i < x.size(); i++) x[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
  cpu.train(2000, x.data());
  cpu.add(2000, x.data());
  printf("reference CPU index: ntotal=%ld use_precomputed_table=%d\n", cpu.ntotal, cpu.use_precomputed_table);
  if (argc > 1) {   // only with a GPU: ./link_against_reference gpu
    faiss::gpu::StandardGpuResources res;
    faiss::gpu::GpuIndexIVFPQ gpu(&res, &cpu);
    gpu.setNumProbes(4);
    std::vector<float> D(10 * 5), D2(10 * 5);
    std::vector<faiss::Index::idx_t> I(10 * 5), I2(10 * 5);
    gpu.search(10, x.data(), 5, D.data(), I.data());
    cpu.nprobe = 4;
    cpu.search(10, x.data(), 5, D2.data(), I2.data());
    printf("gpu vs reference cpu: labels %s\n", I == I2 ? "equal" : "differ");
  }
  return 0;
}
