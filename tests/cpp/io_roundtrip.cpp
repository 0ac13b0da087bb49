// read_index(file written by the reference) -> write_index -> must be byte-identical;
// prints the fields the pytest compares with the fixture.  No GPU needed (precompute=false).
#include <cstdio>
#include "faiss_amd/index_io.h"

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s in.faissindex out.faissindex\n", argv[0]); return 2; }
  faiss::Index* idx = faiss::read_index(argv[1], /*precompute=*/false);
  faiss::IndexIVFPQ* iv = dynamic_cast<faiss::IndexIVFPQ*>(idx);
  if (!iv) { fprintf(stderr, "not an IndexIVFPQ\n"); return 1; }
  size_t nvec = 0, ncodes = 0;
  for (size_t i = 0; i < iv->nlist; i++) { nvec += iv->ids[i].size(); ncodes += iv->codes[i].size(); }
  const bool imi = dynamic_cast<faiss::MultiIndexQuantizer*>(iv->quantizer) != nullptr;
  printf("d=%d ntotal=%ld nlist=%zu nprobe=%zu M=%zu nbits=%zu code_size=%zu by_residual=%d trained=%d "
         "nvec=%zu ncodes=%zu quantizer=%s qntotal=%ld\n",
         iv->d, iv->ntotal, iv->nlist, iv->nprobe, iv->pq.M, iv->pq.nbits, iv->code_size, (int)iv->by_residual,
         (int)iv->is_trained, nvec, ncodes, imi ? "Imiq" : "IxF2", iv->quantizer->ntotal);
  faiss::write_index(idx, argv[2]);
  delete idx;
  return 0;
}
