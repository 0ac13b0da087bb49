#!/usr/bin/env python3
"""Per-probe phase clocks of scan16_kernel on the headline data (diagnostic build of the library):
   tools/build_variant.sh phases "-DVLQ_SCAN16_PHASES" scan16
   VLQ_LIB_PATH=vector_line_quantization_amd/csrc/variants/libvlq_phases.so VLQ_SCAN16_PHASES=1 python tools/scan16_phases.py
DATA=g1 for rounds 1-3's generator setting; NQ / NPROBE / K as in the bench."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
if os.environ.get("DATA") == "g1":
    a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.03, gmm_centres=2000, rank=0, spread=0.0)
else:
    a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.005, gmm_centres=2000, rank=12, spread=0.4)
nq, nprobe, k = int(os.environ.get("NQ", 10000)), int(os.environ.get("NPROBE", 32)), int(os.environ.get("K", 10))
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq = bench.gmm(torch, gen, centres, nq, a.sigma, dev, a.rank, a.spread)
D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
for _ in range(100): g.search(xq, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
g.stats(reset=True)
g.profile(2); g.profile_read(reset=True)
reps = 20
for _ in range(reps): g.search(xq, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
p = g.profile_read(reset=True); g.profile(False)
print("scan kernel %.3f ms per launch (instrumented build: the stamps cost about a tenth)" % (p["scan_ms"] / reps), flush=True)
st = g.stats(reset=True)        # the library prints the phase table to stderr (VLQ_SCAN16_PHASES=1)
print("codes per query %.1f" % (st[1] / (reps * nq)))
