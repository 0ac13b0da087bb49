#!/usr/bin/env python3
"""Generic scan kernel timing (code sizes other than 16 bytes): python tools/time_generic.py M [nbits] [d]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vector_line_quantization_amd as vlq
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nbits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
nq, nlist, nb, nprobe, k = 10000, 4096, 1000000, 32, 10
rng = np.random.default_rng(0)
g = vlq.GpuIVFPQ(d, nlist, M, nbits)
cent = rng.random((nlist, d), dtype=np.float32)
g.set_coarse_centroids(cent)
g.set_pq_centroids(rng.random((M, 1 << nbits, d // M), dtype=np.float32))
lens = rng.multinomial(nb, rng.dirichlet(np.full(nlist, 1.2)))
off = np.zeros(nlist + 1, np.int64); np.cumsum(lens, out=off[1:])
g.set_lists(rng.integers(0, 1 << nbits, (nb, M), dtype=np.uint8), np.arange(nb, dtype=np.int64), off)
x = torch.from_numpy(cent[rng.integers(0, nlist, nq)] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)).float().cuda()
D = torch.empty((nq, k), dtype=torch.float32, device="cuda"); I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
for _ in range(2): g.search(x, nprobe, k, D=D, I=I)
torch.cuda.synchronize(); g.stats(reset=True); g.profile(1); g.profile_read(reset=True)
reps = 5
for _ in range(reps): g.search(x, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
p = g.profile_read(); _n, ncode = g.stats()
print("M=%d nbits=%d d=%d: coarse %.3f tables %.3f scan %.3f ms per 10k queries; %.0f codes/query; scan %.2f TB/s of code bytes" % (
    M, nbits, d, p["coarse_ms"] / reps, p["tables_ms"] / reps, p["scan_ms"] / reps, ncode / reps / nq, ncode / reps * M / (p["scan_ms"] / reps * 1e-3) / 1e12))
