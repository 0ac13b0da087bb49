#!/usr/bin/env python3
"""The clock period a first launch is seeded with (walk_stat_kernel's model) against the period the workgroups then measure,
and the cold search's time against the warm one, over batch sizes / nprobe / k on both bench data sets."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
for name, kw in (("headline", dict(sigma=0.005, rank=12, spread=0.4)), ("G1", dict(sigma=0.03, rank=0, spread=0.0))):
    a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, gmm_centres=2000, **kw)
    g, centres, coarse, pq, xb = bench.build_index(a, dev)
    gen = torch.Generator(device=dev); gen.manual_seed(33)
    xq_all = bench.gmm(torch, gen, centres, 10000, a.sigma, dev, a.rank, a.spread)
    for nq, nprobe, k in ((10000, 32, 10), (10000, 16, 10), (10000, 64, 10), (10000, 128, 10), (5000, 32, 10), (2500, 32, 10), (1250, 32, 10), (10000, 32, 50)):
        xq = xq_all[:nq].contiguous()
        D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        for _ in range(30): g.search(xq, nprobe, k, D=D, I=I)
        torch.cuda.synchronize()
        tc, tw, seeds = [], [], []
        for _ in range(6):
            g.reset_walk_state(); torch.cuda.synchronize()
            t0 = time.perf_counter(); g.search(xq, nprobe, k, D=D, I=I); torch.cuda.synchronize(); tc.append(time.perf_counter() - t0)
            seeds.append(g.last_scan_info().split("launch_period_ticks=")[1])
            for _ in range(3): g.search(xq, nprobe, k, D=D, I=I)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); g.search(xq, nprobe, k, D=D, I=I); torch.cuda.synchronize(); tw.append(time.perf_counter() - t0)
        info = g.last_scan_info()
        tc.sort(); tw.sort()
        print("%-8s nq %5d nprobe %3d k %3d: cold %.3f ms warm %.3f ms (x %.3f); seeded period %s, measured %s; %s" % (
            name, nq, nprobe, k, tc[3] * 1e3, tw[3] * 1e3, tc[3] / tw[3], seeds[-1], info.split("period_ticks=")[1].split()[0],
            info.split(" first=")[0]), flush=True)
