#!/usr/bin/env python3
"""Throughput / latency sweep of the search path on the bench index (one MI355X):
batch size x nprobe x k.  Writes a markdown table.   python tools/sweep.py > profiles/r01_sweep.md"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.03, gmm_centres=2000, rank=0, spread=0.0)
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq_all = bench.gmm(torch, gen, centres, 100000, a.sigma, dev)
print("# Search sweep, one MI355X, bench index (d=128, nlist=4096, M=16x8 bit, 1 M vectors, generator G1)\n")
print("Queries and results resident in HBM; time per `search()` call = median of 7 after 2 warm-ups.\n")
print("| batch | nprobe | k | ms per call | queries/s | codes per query |")
print("|---|---|---|---|---|---|")
for nq in (1, 16, 128, 1000, 1250, 2500, 5000, 10000, 100000):      # 1250 / 2500 / 5000: slices of a 10 000-query batch over 8 / 4 / 2 GPUs
    for nprobe, k in ((32, 10), (8, 10), (128, 10), (32, 1), (32, 100), (32, 1000)):
        if nq == 100000 and (nprobe, k) not in ((32, 10), (8, 10)):
            continue
        if nq in (1250, 2500, 5000) and (nprobe, k) != (32, 10):
            continue
        xq = xq_all[:nq].contiguous()
        D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        for _ in range(2): g.search(xq, nprobe, k, D=D, I=I)
        torch.cuda.synchronize(); g.stats(reset=True)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); g.search(xq, nprobe, k, D=D, I=I); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        _n, ncode = g.stats(reset=True)
        ts.sort(); t = ts[3]
        print("| %d | %d | %d | %.3f | %.0f | %.0f |" % (nq, nprobe, k, t * 1e3, nq / t, ncode / 7 / nq), flush=True)
