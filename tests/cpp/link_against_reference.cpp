// Compile/link check of INTEGRATION.md §A: the GPU shell compiled against the
// REFERENCE's own CPU headers and linked with the reference's own CPU library
// (oracle/_ref/libfaiss_ref.so).  Built only where /root/reference exists (the binary is
// git-ignored and travels to the GPU box like the other built files); without an argument it stops
// after constructing the reference CPU index (no GPU in the build container), with `gpu` it copies
// that index into the GPU shell and compares the answers (tests/test_cpp_shell.py, -m gpu).
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
    // distances bit-identical; labels identical up to exact-distance ties (the CPU heap's history)
    bool dis_eq = D == D2;
    size_t lab_eq = 0;
    for (size_t i = 0; i < I.size(); i++) lab_eq += I[i] == I2[i] || D[i] == D2[i];
    printf("gpu vs reference cpu: distances %s, labels %s\n", dis_eq ? "bit-equal" : "differ",
           lab_eq == I.size() ? "equal" : "differ");
    // and back: the GPU index copied into a reference-class CPU index answers like the original
    faiss::IndexFlatL2 q2(d);
    faiss::IndexIVFPQ back(&q2, d, nlist, 8, 8);
    gpu.copyTo(&back);
    back.nprobe = 4;
    std::vector<float> D3(10 * 5);
    std::vector<faiss::Index::idx_t> I3(10 * 5);
    back.search(10, x.data(), 5, D3.data(), I3.data());
    printf("copyTo -> reference cpu: %s\n", (D3 == D2 && I3 == I2) ? "equal" : "differ");
    return (dis_eq && lab_eq == I.size() && D3 == D2 && I3 == I2) ? 0 : 1;
  }
  return 0;
}
