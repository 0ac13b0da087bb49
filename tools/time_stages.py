#!/usr/bin/env python3
"""Stage timing on random data (kernel experiments): coarse / scan via the library's
HIP-event profile.  python tools/time_stages.py [nq] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vector_line_quantization_amd as vlq

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d, nlist, M = 128, 4096, 16
rng = np.random.default_rng(0)
g = vlq.GpuIVFPQ(d, nlist, M, 8)
g.set_coarse_centroids(rng.random((nlist, d), dtype=np.float32))
g.set_pq_centroids(rng.random((M, 256, d // M), dtype=np.float32))
import torch
x = torch.from_numpy(rng.random((nq, d), dtype=np.float32)).cuda()
cd = torch.empty((nq, 32), dtype=torch.float32, device="cuda")
keys = torch.empty((nq, 32), dtype=torch.int64, device="cuda")
for _ in range(2):
    g.coarse_search(x, 32, cdis=cd, keys=keys)
torch.cuda.synchronize()
g.profile(True); g.profile_read(reset=True)
for _ in range(reps):
    g.coarse_search(x, 32, cdis=cd, keys=keys)
torch.cuda.synchronize()
p = g.profile_read()
print("lib=%s coarse_ms=%.4f" % (os.environ.get("VLQ_LIB_PATH", "default"), p["coarse_ms"] / reps))

# scan stage on a synthetic index with SIFT1M-like list sizes
nb = int(os.environ.get('NB', 1000000))
lens = rng.multinomial(nb, rng.dirichlet(np.full(nlist, 1.2)))
off = np.zeros(nlist + 1, np.int64); np.cumsum(lens, out=off[1:])
g.set_lists(rng.integers(0, 256, (nb, M), dtype=np.uint8), np.arange(nb, dtype=np.int64), off)
D = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
I = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
for _ in range(2):
    g.search_preassigned(x, keys, cd, 10, D=D, I=I)
torch.cuda.synchronize()
g.stats(reset=True); g.profile_read(reset=True)
for _ in range(reps):
    g.search_preassigned(x, keys, cd, 10, D=D, I=I)
torch.cuda.synchronize()
p = g.profile_read()
_, ncode = g.stats()
print("scan_ms=%.4f ncode/query=%.0f -> %.0f GB/s algorithmic" % (p["scan_ms"] / reps, ncode / reps / nq, ncode / reps * 16 / (p["scan_ms"] / reps * 1e-3) / 1e9))
