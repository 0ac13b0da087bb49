#!/usr/bin/env python3
"""One schedule of the 16-byte scan on the bench's headline data (recall-conformant generator setting), scan-only timing
and, for builds with -DVLQ_PHASE_TIMING (make FLAGS_scan16o=-DVLQ_PHASE_TIMING), the per-workgroup phase clocks.
   MODE=3 python tools/owned_probe.py [reps]      env: MODE (1 query-major, 2 / 3 / 4 list-owned builds), K, NQ"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
a = argparse.Namespace(nq=int(os.environ.get("NQ", 10000)), nb=1000000, nt=100000, d=128, nlist=4096, M=16, nprobe=32,
                       k=int(os.environ.get("K", 10)), sigma=0.005, gmm_centres=2000, rank=12, spread=0.4)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq = bench.gmm(torch, gen, centres, a.nq, a.sigma, dev, a.rank, a.spread)
D = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev); I = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
for mode in [int(v) for v in os.environ.get("MODE", "1,3").split(",")]:
    g.set_scan_schedule(mode)
    for _ in range(30): g.search(xq, a.nprobe, a.k, D=D, I=I)
    torch.cuda.synchronize()
    g.stats(reset=True); g.profile(1); g.profile_read(reset=True)
    for _ in range(reps): g.search(xq, a.nprobe, a.k, D=D, I=I)
    torch.cuda.synchronize()
    p = g.profile_read(reset=True); g.profile(False)
    print("schedule %d: coarse %.3f tables %.3f scan %.3f ms" % (mode, p["coarse_ms"] / reps, p["tables_ms"] / reps, p["scan_ms"] / reps), flush=True)
    g.stats(reset=True)
