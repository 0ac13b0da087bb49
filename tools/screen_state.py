#!/usr/bin/env python3
"""Rows the coarse screen could not decide on the bench data (both generator settings), and the coarse stage's time."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for name, kw in (("headline", dict(sigma=0.005, rank=12, spread=0.4)), ("G1", dict(sigma=0.03, rank=0, spread=0.0))):
    a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, gmm_centres=2000, **kw)
    dev = torch.device("cuda", 0)
    g, centres, coarse, pq, xb = bench.build_index(a, dev)
    gen = torch.Generator(device=dev); gen.manual_seed(33)
    xq = bench.gmm(torch, gen, centres, 10000, a.sigma, dev, a.rank, a.spread)
    for nprobe in (16, 32, 64):
        en0, rows0, und0 = g.coarse_screen_state()
        cd, keys = g.coarse_search(xq, nprobe)
        torch.cuda.synchronize()
        cd, keys = g.coarse_search(xq, nprobe)
        torch.cuda.synchronize()
        en, rows, und = g.coarse_screen_state()
        print("%s nprobe %d: screen enabled %s, rows %d, undecided %d" % (name, nprobe, en, rows - rows0, und - und0), flush=True)
