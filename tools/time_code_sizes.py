#!/usr/bin/env python3
"""Scan timing over the code sizes (M x 8 bit) on the bench's two generator settings: the headline data (recall-conformant,
~330 codes per visited list) and G1 (sigma 0.03, ~700 codes per visited list), 10 000 queries, nprobe 32, k 10.  Per size:
stage times from the library's HIP events, and the scan's rate as a fraction of the LDS gather ceiling (one random 4-byte
LDS gather per code byte: 9.98 lanes per clock and CU, tools/micro/lds_gather.hip) and of the HBM peak (code bytes).
   python tools/time_code_sizes.py [reps]      env: MS=4,8,...,64 (default: every engineered size)  SETS=headline,G1  GENERIC=1 (the generic kernel, via VLQ_GENERIC_SCAN)"""
import argparse, copy, os, sys
if os.environ.get("GENERIC"):
    os.environ["VLQ_GENERIC_SCAN"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
base = argparse.Namespace(nq=10000, nb=1000000, nt=100000, d=128, nlist=4096, M=16, nprobe=32, k=10, sigma=0.005,
                          gmm_centres=2000, rank=12, spread=0.4)
for name, kw in (("headline", {}), ("G1", dict(sigma=0.03, rank=0, spread=0.0))):
    if name not in os.environ.get("SETS", "headline,G1").split(","):
        continue
    for M in [int(v) for v in os.environ.get("MS", "4,8,12,16,20,24,28,32,40,48,56,64").split(",")]:
        a = copy.copy(base)
        a.M = M
        a.d = 128 if 128 % M == 0 else M * (128 // M)        # 12 / 20 / 24 / 28 / 40 / 48 / 56 bytes: d = 120 / 120 / 120 / 112 / 120 / 96 / 112
        for k_, v in kw.items():
            setattr(a, k_, v)
        g, centres, coarse, pq, xb = bench.build_index(a, dev)
        gen = torch.Generator(device=dev); gen.manual_seed(33)
        xq = bench.gmm(torch, gen, centres, a.nq, a.sigma, dev, a.rank, a.spread)
        D = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev); I = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
        for _ in range(30): g.search(xq, a.nprobe, a.k, D=D, I=I)
        torch.cuda.synchronize()
        g.stats(reset=True); g.profile(1); g.profile_read(reset=True)
        for _ in range(reps): g.search(xq, a.nprobe, a.k, D=D, I=I)
        torch.cuda.synchronize()
        p = g.profile_read(reset=True); g.profile(False)
        _n, ncode = g.stats(reset=True)
        ncode /= reps
        scan = p["scan_ms"] / reps
        gather = ncode * M / (scan * 1e-3) / 256 / 2.4e9
        print("%-8s M=%-2d (%d-byte codes)%s: coarse %.3f tables %.3f scan %.3f ms; %.0f codes/query; code bytes %.2f TB/s = %.2f of HBM peak; "
              "%.2f gathers per clock and CU = %.2f of the LDS gather ceiling" % (
                  name, M, M, " GENERIC KERNEL" if os.environ.get("GENERIC") and M != 16 else "", p["coarse_ms"] / reps, p["tables_ms"] / reps, scan,
                  ncode / a.nq, ncode * M / (scan * 1e-3) / 1e12, ncode * M / (scan * 1e-3) / 8e12, gather, gather / 9.98), flush=True)
        del g
