#!/usr/bin/env python3
"""Phase breakdown of the wave-autonomous scan (kernel experiments): needs the stamped
build `make -C vector_line_quantization_amd/csrc libvlq_stamps.so`.
   VLQ_LIB_PATH=.../libvlq_stamps.so VLQ_SCAN16=w4 python tools/stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vector_line_quantization_amd as vlq
from vector_line_quantization_amd._lib import lib

nq, d, nlist, M = 10000, 128, 4096, 16
rng = np.random.default_rng(0)
g = vlq.GpuIVFPQ(d, nlist, M, 8)
g.set_coarse_centroids(rng.random((nlist, d), dtype=np.float32))
g.set_pq_centroids(rng.random((M, 256, d // M), dtype=np.float32))
x = torch.from_numpy(rng.random((nq, d), dtype=np.float32)).cuda()
cd = torch.empty((nq, 32), dtype=torch.float32, device="cuda")
keys = torch.empty((nq, 32), dtype=torch.int64, device="cuda")
g.coarse_search(x, 32, cdis=cd, keys=keys)
nb = int(os.environ.get("NB", 3000000))
lens = rng.multinomial(nb, rng.dirichlet(np.full(nlist, 1.2)))
off = np.zeros(nlist + 1, np.int64); np.cumsum(lens, out=off[1:])
g.set_lists(rng.integers(0, 256, (nb, M), dtype=np.uint8), np.arange(nb, dtype=np.int64), off)
D = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
I = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
for _ in range(2):
    g.search_preassigned(x, keys, cd, 10, D=D, I=I)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
L = lib()
L.vlq_debug_stamps(out, 1)
g.profile(True); g.profile_read(reset=True)
g.search_preassigned(x, keys, cd, 10, D=D, I=I)
torch.cuda.synchronize()
p = g.profile_read()
L.vlq_debug_stamps(out, 0)
v = list(out)
nw = v[8]
names = ["prologue", "wait_vm", "build", "claim+prefetch", "scan", "iters", "probe_loop", "epilogue"]
print("scan_ms=%.3f waves=%d" % (p["scan_ms"], nw))
for i, n in enumerate(names):
    print("%-15s %12.0f per wave" % (n, v[i] / max(nw, 1)))
print("scan ticks per iteration: %.1f  (code wait %.1f, gather+add %.1f, offer %.1f)" % (v[4] / max(v[5], 1), v[9] / max(v[5], 1), v[10] / max(v[5], 1), v[11] / max(v[5], 1)))
