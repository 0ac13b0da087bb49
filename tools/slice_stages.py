#!/usr/bin/env python3
"""Stage times of the slices a 10 000-query batch leaves per GPU under strong scaling (bench index)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
# DATA=g1: rounds 1-3's generator setting (sigma 0.03 isotropic); default: bench.py's headline setting
if os.environ.get("DATA") == "g1":
    a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.03, gmm_centres=2000, rank=0, spread=0.0)
else:
    a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.005, gmm_centres=2000, rank=12, spread=0.4)
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq_all = bench.gmm(torch, gen, centres, 10000, a.sigma, dev, a.rank, a.spread)
for nq in (625, 1000, 1024, 1100, 1250, 1500, 2048, 2500, 5000, 10000):
    xq = xq_all[:nq].contiguous()
    D = torch.empty((nq, 10), dtype=torch.float32, device=dev); I = torch.empty((nq, 10), dtype=torch.int64, device=dev)
    for _ in range(3): g.search(xq, 32, 10, D=D, I=I)
    torch.cuda.synchronize(); g.profile(1); g.profile_read(reset=True)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps): g.search(xq, 32, 10, D=D, I=I)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    p = g.profile_read(reset=True); g.profile(False)
    t0 = time.perf_counter()
    for _ in range(reps): g.search(xq, 32, 10, D=D, I=I)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / reps
    print("nq %5d: %.3f ms per call uninstrumented (%.2f M q/s; x8 = %.1f M); coarse %.3f tables %.3f scan %.3f" % (
        nq, dt2 * 1e3, nq / dt2 / 1e6, 8 * nq / dt2 / 1e6 if nq == 1250 else nq / dt2 / 1e6, p["coarse_ms"] / reps, p["tables_ms"] / reps, p["scan_ms"] / reps), flush=True)
