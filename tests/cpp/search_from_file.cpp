// A reference-written index file -> the MI355X -> answers.  read_index() of a file produced by the
// REFERENCE's faiss::write_index (index_io.cpp:240-355; the bytes are stored as data in the golden
// fixtures), then (a) the IndexIVFPQ object itself searches (its lists are mirrored to HBM), and
// (b) for a flat-L2 coarse quantizer the index is copyFrom'ed into a GpuIndexIVFPQ
// (gpu/GpuIndexIVFPQ.cu:168-231) which searches again.  Distances and labels of both are written as raw
// little-endian arrays; tests/test_index_io.py compares them with the fixture's reference results.
//   search_from_file index.faissindex queries.f32 nq nprobe k out_prefix
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "faiss_amd/gpu/GpuIndexIVFPQ.h"
#include "faiss_amd/gpu/StandardGpuResources.h"
#include "faiss_amd/index_io.h"

static bool dump(const std::string& fn, const void* p, size_t bytes) {
  FILE* f = fopen(fn.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(p, 1, bytes, f) == bytes;
  fclose(f);
  return ok;
}

int main(int argc, char** argv) {
  if (argc != 7) { fprintf(stderr, "usage: %s index queries.f32 nq nprobe k out_prefix\n", argv[0]); return 2; }
  const long nq = atol(argv[3]), nprobe = atol(argv[4]), k = atol(argv[5]);
  const std::string out = argv[6];
  faiss::Index* idx = faiss::read_index(argv[1]);
  faiss::IndexIVFPQ* iv = dynamic_cast<faiss::IndexIVFPQ*>(idx);
  if (!iv) { fprintf(stderr, "not an IndexIVFPQ\n"); return 1; }
  std::vector<float> xq((size_t)nq * iv->d);
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(xq.data(), sizeof(float), xq.size(), f) != xq.size()) { fprintf(stderr, "cannot read the queries\n"); return 1; }
  fclose(f);
  iv->nprobe = nprobe;
  std::vector<float> D((size_t)nq * k);
  std::vector<faiss::Index::idx_t> I((size_t)nq * k);
  faiss::indexIVFPQ_stats.reset();
  iv->search(nq, xq.data(), k, D.data(), I.data());
  if (!dump(out + ".D", D.data(), D.size() * 4) || !dump(out + ".I", I.data(), I.size() * 8)) return 1;
  printf("ivfpq: ntotal=%ld nlist=%zu use_precomputed_table=%d ncode=%zu\n", iv->ntotal, iv->nlist,
         iv->use_precomputed_table, faiss::indexIVFPQ_stats.ncode);
  if (dynamic_cast<faiss::IndexFlat*>(iv->quantizer)) {
    faiss::gpu::StandardGpuResources res;
    faiss::gpu::GpuIndexIVFPQConfig config;
    config.usePrecomputedTables = true;
    faiss::gpu::GpuIndexIVFPQ gpu(&res, iv, config);
    gpu.setNumProbes((int)nprobe);
    gpu.search(nq, xq.data(), k, D.data(), I.data());
    if (!dump(out + ".gpu.D", D.data(), D.size() * 4) || !dump(out + ".gpu.I", I.data(), I.size() * 8)) return 1;
    printf("gpu: copyFrom ntotal=%ld lists=%d\n", gpu.ntotal, gpu.getNumLists());
  }
  delete idx;
  return 0;
}
