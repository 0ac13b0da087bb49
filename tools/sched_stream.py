#!/usr/bin/env python3
"""A/B of the query-major kernel (schedule 1) against the persistent stream kernel (3: overlapped, 4: serial)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np, torch
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
base = argparse.Namespace(nq=10000, nb=1000000, nt=100000, d=128, nlist=4096, M=16, nprobe=32, k=10, sigma=0.03,
                          gmm_centres=2000, rank=0, spread=0.0)
for name, kw in (("G1", {}), ("second", dict(sigma=0.005, rank=12, spread=0.4))):
    a = copy.copy(base)
    for k_, v in kw.items(): setattr(a, k_, v)
    g, centres, coarse, pq, xb = bench.build_index(a, dev)
    gen = torch.Generator(device=dev); gen.manual_seed(33)
    xq = bench.gmm(torch, gen, centres, a.nq, a.sigma, dev, a.rank, a.spread)
    ref = None
    for k in (10, 64):
        for mode in (1, 3, 4):
            g.set_scan_schedule(mode)
            Dk = torch.empty((a.nq, k), dtype=torch.float32, device=dev); Ik = torch.empty((a.nq, k), dtype=torch.int64, device=dev)
            for _ in range(3): g.search(xq, a.nprobe, k, D=Dk, I=Ik)
            torch.cuda.synchronize()
            g.stats(reset=True); g.profile(2); g.profile_read(reset=True)
            for _ in range(reps): g.search(xq, a.nprobe, k, D=Dk, I=Ik)
            torch.cuda.synchronize()
            p2 = g.profile_read(reset=True); g.profile(False)
            _n, ncode = g.stats(reset=True)
            same = ""
            if mode == 1: ref = (Dk.clone(), Ik.clone(), ncode)
            else: same = " equal to schedule 1: %s, ncode equal: %s" % (bool(torch.equal(ref[0], Dk) and torch.equal(ref[1], Ik)), ncode == ref[2])
            print("%-6s k=%-3d schedule %d: scan %.3f ms%s" % (name, k, mode, p2["scan_ms"] / max(1, p2["scan_calls"]), same), flush=True)
    del g
