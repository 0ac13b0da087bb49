#!/usr/bin/env python3
"""The coarse stage alone on the bench data: 40 calls of coarse_search(10 000 queries, nprobe) -- for rocprofv3 --kernel-trace --stats."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, gmm_centres=2000, sigma=0.005, rank=12, spread=0.4)
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
nq = int(os.environ.get("NQ", 10000))
xq = bench.gmm(torch, gen, centres, nq, a.sigma, dev, a.rank, a.spread)
nprobe = int(os.environ.get("NPROBE", 32))
for _ in range(40): g.coarse_search(xq, nprobe)
torch.cuda.synchronize()
