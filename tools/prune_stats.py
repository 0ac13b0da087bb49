#!/usr/bin/env python3
"""How many (query, probe) pairs could be skipped by the exact lower bound
dis0 + sum_m min_j T[m][j] >= k-th best distance?  (kernel experiments; bench data)"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=float(os.environ.get("SIGMA", 0.03)),
                          gmm_centres=2000, rank=int(os.environ.get("RANK_", 0)), spread=float(os.environ.get("SPREAD", 0.0)))
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq = bench.gmm(torch, gen, centres, 10000, a.sigma, dev, a.rank, a.spread)
nq = 300
x = xq[:nq]
D, I = g.search(x, 32, 10)
cd, keys = g.coarse_search(x, 32)
D = torch.as_tensor(D).cpu().numpy() if not isinstance(D, np.ndarray) else D
cd = torch.as_tensor(cd).cpu().numpy(); keys = torch.as_tensor(keys).cpu().numpy()
pqc = pq.cpu().numpy(); co = coarse.cpu().numpy(); xh = x.cpu().numpy()
lens = np.array([g.list_length(i) for i in range(a.nlist)])
skip_final = 0; tot = 0; codes_skip = 0; codes_tot = 0
by_rank = np.zeros(32); 
for qi in range(nq):
    for p in range(32):
        c = keys[qi, p]
        r = (xh[qi] - co[c]).reshape(16, 8)
        dist = ((r[:, None, :] - pqc) ** 2).sum(2)          # [16][256] = T entries (up to rounding)
        lb = dist.min(1).sum()
        tot += 1; codes_tot += lens[c]
        if lb >= D[qi, 9]:
            skip_final += 1; codes_skip += lens[c]; by_rank[p] += 1
print("pairs skippable with the FINAL 10th distance as threshold: %.3f; codes: %.3f" % (skip_final / tot, codes_skip / codes_tot))
print("by probe rank:", np.round(by_rank / nq, 2))
