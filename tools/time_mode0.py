#!/usr/bin/env python3
"""Table mode 0 (by_residual without the precomputed table: GpuIndexIVFPQConfig::usePrecomputedTables = false, the reference GPU
class's default; IndexIVFPQ.cpp:636-637 on the CPU) against mode 1 on the bench index.   python tools/time_mode0.py [reps]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
a = argparse.Namespace(nq=10000, nb=1000000, nt=100000, d=128, nlist=4096, M=int(os.environ.get("M", 16)), nprobe=32, k=10, sigma=0.005,
                       gmm_centres=2000, rank=12, spread=0.4)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq = bench.gmm(torch, gen, centres, a.nq, a.sigma, dev, a.rank, a.spread)
D = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev); I = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
for mode in (1, 0):
    g.set_search_options(by_residual=True, use_precomputed_table=mode)
    for _ in range(3): g.search(xq, a.nprobe, a.k, D=D, I=I)
    torch.cuda.synchronize()
    g.stats(reset=True); g.profile(1); g.profile_read(reset=True)
    for _ in range(reps): g.search(xq, a.nprobe, a.k, D=D, I=I)
    torch.cuda.synchronize()
    p = g.profile_read(reset=True); g.profile(False)
    print("M=%d use_precomputed_table=%d: coarse %.3f tables %.3f scan %.3f ms per 10 000 queries" % (a.M, mode, p["coarse_ms"] / reps, p["tables_ms"] / reps, p["scan_ms"] / reps), flush=True)
