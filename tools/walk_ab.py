#!/usr/bin/env python3
"""Scan time under the two walking orders of a query's probes (walk_order.cuh): run once with VLQ_WALK_FIRST=-1 (coarse-distance
order) and once with VLQ_WALK_FIRST=1 (nearest probe first, the rest by list id), or without the variable (the library's rule).
   VLQ_WALK_FIRST=1 python tools/walk_ab.py"""
import argparse, copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
base = argparse.Namespace(nq=10000, nb=1000000, nt=100000, d=128, nlist=4096, M=int(os.environ.get("M", 16)), nprobe=32, k=10, sigma=0.005,
                          gmm_centres=2000, rank=12, spread=0.4)
for name, kw in (("headline", {}), ("G1", dict(sigma=0.03, rank=0, spread=0.0))):
    a = copy.copy(base)
    for k_, v in kw.items():
        setattr(a, k_, v)
    g, centres, coarse, pq, xb = bench.build_index(a, dev)
    gen = torch.Generator(device=dev); gen.manual_seed(33)
    xq_all = bench.gmm(torch, gen, centres, a.nq, a.sigma, dev, a.rank, a.spread)
    cases = ((0, 32, 10, 10000), (0, 32, 100, 10000), (0, 32, 200, 10000), (1, 32, 10, 10000), (0, 8, 10, 10000), (0, 128, 10, 10000),
             (0, 32, 10, 5000), (0, 32, 10, 2500), (0, 32, 10, 1250))
    if os.environ.get("CASES"):
        cases = tuple(tuple(int(v) for v in c.split(",")) for c in os.environ["CASES"].split(";"))
    for fp16, nprobe, k, nq in cases:
        xq = (xq_all / 256 if fp16 else xq_all)[:nq].contiguous()
        if fp16:
            # the half range needs scaled data (bench.py's float16 leg): a second index on the scaled vectors
            continue
        D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        for _ in range(20): g.search(xq, nprobe, k, D=D, I=I)
        torch.cuda.synchronize(); g.profile(1); g.profile_read(reset=True)
        reps = 20
        for _ in range(reps): g.search(xq, nprobe, k, D=D, I=I)
        torch.cuda.synchronize()
        p = g.profile_read(reset=True); g.profile(False)
        print("%-8s nq %5d nprobe %3d k %3d: scan %.3f ms (coarse %.3f, order %.3f)" % (name, nq, nprobe, k, p["scan_ms"] / reps, p["coarse_ms"] / reps, p["tables_ms"] / reps), flush=True)
