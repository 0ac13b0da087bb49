#!/usr/bin/env python3
"""Per-step times of the bench batch right after set-up and after a 0.5 s pause: how long the GPU takes to reach its
clocks (why bench.py warms up for 40 steps).  MI355X, round 3: 0.89 0.83 0.79 0.76 0.76 ... ms per step for consecutive
groups of ten; with a synchronisation after every step 0.98 -> 0.78 over ~35 steps."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.03, gmm_centres=2000, rank=0, spread=0.0)
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq = bench.gmm(torch, gen, centres, 10000, a.sigma, dev)
D = torch.empty((10000, 10), dtype=torch.float32, device=dev); I = torch.empty((10000, 10), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for trial in range(2):
    time.sleep(0.5 if trial else 0.0)
    ts = []
    for i in range(60):
        t0 = time.perf_counter(); g.search(xq, 32, 10, D=D, I=I); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("trial", trial, " ".join("%.3f" % t for t in ts))
# back-to-back groups of 10 without per-step sync
time.sleep(0.5)
gs = []
for grp in range(8):
    t0 = time.perf_counter()
    for i in range(10): g.search(xq, 32, 10, D=D, I=I)
    torch.cuda.synchronize(); gs.append((time.perf_counter() - t0) * 100)
print("groups of 10 (ms per step):", " ".join("%.3f" % t for t in gs))
