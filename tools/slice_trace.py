#!/usr/bin/env python3
"""One batch size of the bench index under rocprofv3 --kernel-trace --stats: which kernels a search of NQ queries launches.
   NQ=1250 rocprofv3 --kernel-trace --stats -- python tools/slice_trace.py"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.005, gmm_centres=2000, rank=12, spread=0.4)
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
nq = int(os.environ.get("NQ", 1250))
xq = bench.gmm(torch, gen, centres, nq, a.sigma, dev, a.rank, a.spread)
D = torch.empty((nq, 10), dtype=torch.float32, device=dev); I = torch.empty((nq, 10), dtype=torch.int64, device=dev)
for _ in range(200): g.search(xq, 32, 10, D=D, I=I)
torch.cuda.synchronize()
