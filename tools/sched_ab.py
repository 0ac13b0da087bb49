#!/usr/bin/env python3
"""A/B of the two scan schedules (query-major / list-owned, include/vlq_ivfpq.h) on the bench's two
data sets: per-stage times from the library's HIP events, same box, same index.
   python tools/sched_ab.py [reps]"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np, torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
base = argparse.Namespace(nq=10000, nb=1000000, nt=100000, d=128, nlist=4096, M=16, nprobe=32, k=10, sigma=0.03,
                          gmm_centres=2000, rank=0, spread=0.0)
MODES = [int(v) for v in os.environ.get("MODES", "1,2,3,4").split(",")]
for name, kw in (("G1", {}), ("second", dict(sigma=0.005, rank=12, spread=0.4))):
    a = copy.copy(base)
    for k_, v in kw.items():
        setattr(a, k_, v)
    g, centres, coarse, pq, xb = bench.build_index(a, dev)
    gen = torch.Generator(device=dev); gen.manual_seed(33)
    xq = bench.gmm(torch, gen, centres, a.nq, a.sigma, dev, a.rank, a.spread)
    D = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev); I = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
    ref = None
    for k in (a.k, 100):
        for mode in MODES:
            g.set_scan_schedule(mode)
            Dk = torch.empty((a.nq, k), dtype=torch.float32, device=dev); Ik = torch.empty((a.nq, k), dtype=torch.int64, device=dev)
            for _ in range(3): g.search(xq, a.nprobe, k, D=Dk, I=Ik)
            torch.cuda.synchronize()
            g.stats(reset=True); g.profile(1); g.profile_read(reset=True)
            t0 = time.perf_counter()
            for _ in range(reps): g.search(xq, a.nprobe, k, D=Dk, I=Ik)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            p = g.profile_read(reset=True); g.profile(False)
            g.profile(2); g.profile_read(reset=True)
            for _ in range(reps): g.search(xq, a.nprobe, k, D=Dk, I=Ik)
            torch.cuda.synchronize()
            p2 = g.profile_read(reset=True); g.profile(False)
            same = ""
            if mode == 1: ref = (Dk.clone(), Ik.clone())
            else: same = " results equal query-major: %s" % bool(torch.equal(ref[0], Dk) and torch.equal(ref[1], Ik))
            print("%-6s k=%-3d schedule %d: step %.3f ms (instrumented); coarse %.3f tables %.3f scan %.3f | scan-only timing %.3f ms%s" % (
                name, k, mode, dt * 1e3, p["coarse_ms"] / reps, p["tables_ms"] / reps, p["scan_ms"] / reps, p2["scan_ms"] / max(1, p2["scan_calls"]), same), flush=True)
    del g
