#!/usr/bin/env python3
"""The multi-index coarse stage alone (no lists): coarse_search(nq queries, nprobe) on 2 x NBITS bits, d = 128.
   python tools/time_imi_coarse.py   env: NBITS (14), NPROBES ("64,2048"), NQS ("256,1280,2560,10000"), REPS"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vector_line_quantization_amd as vlq
E = lambda k, v: os.environ.get(k, v)
nbits, d, M = int(E("NBITS", "14")), int(E("D", "128")), 16
rng = np.random.default_rng(0)
g = vlq.GpuIVFPQ(d, 1 << (2 * nbits), M, 8)
g.set_stream(torch.cuda.current_stream().cuda_stream)
g.set_imi_centroids(nbits, rng.random((2, 1 << nbits, d // 2), dtype=np.float32))
g.set_pq_centroids(rng.random((M, 256, d // M), dtype=np.float32))
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
reps = int(E("REPS", "5"))
for nprobe in [int(v) for v in E("NPROBES", "64,2048").split(",")]:
    for nq in [int(v) for v in E("NQS", "256,1280,2560,10000").split(",")]:
        xq = torch.rand((nq, d), device="cuda", generator=gen)
        cd = torch.empty((nq, nprobe), dtype=torch.float32, device="cuda"); keys = torch.empty((nq, nprobe), dtype=torch.int64, device="cuda")
        for _ in range(2): g.coarse_search(xq, nprobe, cdis=cd, keys=keys)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(reps): g.coarse_search(xq, nprobe, cdis=cd, keys=keys)
        torch.cuda.synchronize(); dt = (time.time() - t0) / reps
        print("nprobe %5d  nq %6d: %9.3f ms  (%.2f us per query)" % (nprobe, nq, dt * 1e3, dt * 1e6 / nq), flush=True)
