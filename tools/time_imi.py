#!/usr/bin/env python3
"""Inverted-multi-index configuration timing (kernel experiments): IMI 2 x NBITS, table type 2.
   python tools/time_imi.py [nq]   env: NBITS (default 10), NB, BATCH, NPROBE, K"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vector_line_quantization_amd as vlq
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
E = lambda k, v: int(os.environ.get(k, v))
nbits, nb, nprobe, k, d, M = E("NBITS", 10), E("NB", 4000000), E("NPROBE", 64), E("K", 10), 128, 16
nlist = 1 << (2 * nbits)
rng = np.random.default_rng(0)
g = vlq.GpuIVFPQ(d, nlist, M, 8)
g.set_stream(torch.cuda.current_stream().cuda_stream)   # the library must run in order with torch's generators
imi = rng.random((2, 1 << nbits, d // 2), dtype=np.float32)
g.set_imi_centroids(nbits, imi)
g.set_pq_centroids((rng.random((M, 256, d // M), dtype=np.float32) - 0.5) * 0.2)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
t0 = time.time()
step = E("BATCH", 1000000)
for i in range(0, nb, step):
    n = min(step, nb - i)
    g.add(torch.rand((n, d), device="cuda", generator=gen))
    if (i // step) % 10 == 9:
        torch.cuda.synchronize(); print("  added %d M in %.1f s" % ((i + n) // 1000000, time.time() - t0), flush=True)
torch.cuda.synchronize()
print("added %d vectors into %d lists in %.1f s, device memory in use %.1f GB" % (nb, nlist, time.time() - t0, (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)
xq = torch.rand((nq, d), device="cuda", generator=gen)
D = torch.empty((nq, k), dtype=torch.float32, device="cuda"); I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
for _ in range(2): g.search(xq, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
g.stats(reset=True); g.profile(True); g.profile_read(reset=True)
t0 = time.time(); reps = 5
for _ in range(reps): g.search(xq, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
dt = (time.time() - t0) / reps
p = g.profile_read(); _, ncode = g.stats()
print("search %.3f ms per %d queries = %.0f QPS; stages ms: coarse %.3f tables %.3f scan %.3f; ncode/query %.0f" % (
    dt * 1e3, nq, nq / dt, p["coarse_ms"] / reps, p["tables_ms"] / reps, p["scan_ms"] / reps, ncode / reps / nq))
