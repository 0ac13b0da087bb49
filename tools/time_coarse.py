#!/usr/bin/env python3
"""The coarse stage alone (f32 MFMA matrix + select with the float16 screen off): wall time per call; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.   NQ / NLIST / DIM / NPROBE from the environment
(defaults = BASELINE C1: 10 000 x 4096 x 128, nprobe 32;  the VLQ drivers' shape: NQ=2000 NLIST=65536 DIM=96 NPROBE=64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vector_line_quantization_amd as vlq

nq, nlist, d, nprobe = (int(os.environ.get(k, v)) for k, v in (("NQ", 10000), ("NLIST", 4096), ("DIM", 128), ("NPROBE", 32)))
reps = int(os.environ.get("REPS", 30))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(5)
cent = torch.rand((nlist, d), generator=gen, device=dev)
xq = torch.rand((nq, d), generator=gen, device=dev)
g = vlq.GpuIVFPQ(d, nlist, 16 if d % 16 == 0 else 8, 8)
g.set_coarse_centroids(cent)
g.set_coarse_screen(int(os.environ.get("SCREEN", 0)))
cdis = torch.empty((nq, nprobe), dtype=torch.float32, device=dev)
keys = torch.empty((nq, nprobe), dtype=torch.int64, device=dev)
for _ in range(30):
    g.coarse_search(xq, nprobe, cdis=cdis, keys=keys)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    g.coarse_search(xq, nprobe, cdis=cdis, keys=keys)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fl = 2.0 * nq * nlist * d
print("coarse %d x %d x %d nprobe %d: %.1f us per call (matrix alone would be %.1f TFLOP/s f32 at this time)" % (nq, nlist, d, nprobe, dt * 1e6, fl / dt / 1e12))
# spot check against torch on a few rows (ordering only; bit-exactness is the test suite's job)
ref = torch.cdist(xq[:8], cent).pow(2).topk(nprobe, largest=False).indices.sort(dim=1).values
got = keys[:8].sort(dim=1).values
print("top-%d sets equal on 8 rows: %s" % (nprobe, bool((ref == got).float().mean() > 0.98)))
