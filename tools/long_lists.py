#!/usr/bin/env python3
"""The scan kernel where code bytes dominate: long lists (kernel experiments).
   python tools/long_lists.py [nb] [nlist] [nq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vector_line_quantization_amd as vlq
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64000000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
d, M, nprobe, k = int(os.environ.get("DIM", "128")), 16, int(os.environ.get("NPROBE", "32")), int(os.environ.get("K", "10"))
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
g = vlq.GpuIVFPQ(d, nlist, M, 8)
g.set_stream(torch.cuda.current_stream().cuda_stream)   # the library must run in order with torch's generators
cent = rng.random((nlist, d), dtype=np.float32)
g.set_coarse_centroids(cent)
g.set_pq_centroids(rng.random((M, 256, d // M), dtype=np.float32))
# random codes straight into the lists (the scan does not care what they encode)
lens = rng.multinomial(nb, np.full(nlist, 1.0 / nlist))
off = np.zeros(nlist + 1, np.int64); np.cumsum(lens, out=off[1:])
codes = torch.randint(0, 256, (nb, M), dtype=torch.uint8, device=dev)
ids = torch.arange(nb, dtype=torch.int64, device=dev)
g.set_lists(codes, ids, off)
del codes, ids
x = torch.from_numpy(cent[rng.integers(0, nlist, nq)] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)).float().to(dev)
D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
for _ in range(2): g.search(x, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
g.stats(reset=True); g.profile(2); g.profile_read(reset=True)
reps = 5
for _ in range(reps): g.search(x, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
p = g.profile_read(); _n, ncode = g.stats()
scan_ms = p["scan_ms"] / reps
print("nb=%d nlist=%d (%.0f codes per list) nq=%d: scan kernel %.3f ms, %.0f codes per query, %.2f TB/s of code bytes = %.2f of 8 TB/s" % (
    nb, nlist, nb / nlist, nq, scan_ms, ncode / reps / nq, ncode / reps * 16 / (scan_ms * 1e-3) / 1e12, ncode / reps * 16 / (scan_ms * 1e-3) / 8e12))
g.profile(1); g.profile_read(reset=True)
t0 = time.time()
for _ in range(reps): g.search(x, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
wall = (time.time() - t0) / reps * 1e3
p = g.profile_read()
print("whole search %.3f ms per batch (%.2f M queries/s): coarse %.3f, tables+order %.3f, scan %.3f ms" % (
    wall, nq / wall / 1e3, p["coarse_ms"] / reps, p["tables_ms"] / reps, p["scan_ms"] / reps))
