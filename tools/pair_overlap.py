#!/usr/bin/env python3
"""How many of a query's probe lists does the NEXT query of the scan order probe too?  (Bench index and the second
data set; decides whether two queries per workgroup sharing their LDS gathers can pay.)
   python tools/pair_overlap.py"""
import os, sys, types, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
base = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.03, gmm_centres=2000, rank=0, spread=0.0)
for name, kw in (("bench data", {}), ("second data set", dict(sigma=0.005, rank=12, spread=0.4))):
    a = copy.copy(base)
    for k, v in kw.items(): setattr(a, k, v)
    g, centres, coarse, pq, xb = bench.build_index(a, dev)
    gen = torch.Generator(device=dev); gen.manual_seed(33)
    xq = bench.gmm(torch, gen, centres, 10000, a.sigma, dev, a.rank, a.spread)
    cd, keys = g.coarse_search(xq.cpu().numpy(), 32)
    lens = np.array([g.list_length(i) for i in range(a.nlist)], np.int64)
    def report(order, label):
        A, B = keys[order[0::2]], keys[order[1::2]]
        shared_codes = tot = 0
        sh = []
        for ka, kb in zip(A, B):
            s = np.intersect1d(ka, kb)
            sh.append(len(s))
            shared_codes += 2 * lens[s].sum(); tot += lens[ka].sum() + lens[kb].sum()
        sh = np.array(sh)
        print("%s, %s: shared probes per pair mean %.1f of 32 (min %d, median %d); codes scanned on shared lists %.3f of all" % (
            name, label, sh.mean(), sh.min(), np.median(sh), shared_codes / tot), flush=True)
    report(np.lexsort((keys[:, 1], keys[:, 0])), "pairs = neighbours in (1st, 2nd) nearest-list order")
    # greedy: inside a bin of equal nearest list, pair by sorted probe-set signature
    sig = np.sort(keys, axis=1)
    report(np.lexsort(tuple(sig[:, c] for c in range(7, -1, -1)) + (keys[:, 0],)), "pairs = neighbours by sorted probe set inside the nearest-list bin")
