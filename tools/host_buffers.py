#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) step against the device-resident step on the bench index.
   python tools/host_buffers.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.03, gmm_centres=2000, rank=0, spread=0.0)
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq = bench.gmm(torch, gen, centres, 10000, a.sigma, dev)
xh = xq.cpu().numpy()
D = torch.empty((10000, 10), dtype=torch.float32, device=dev); I = torch.empty((10000, 10), dtype=torch.int64, device=dev)
Dh = np.empty((10000, 10), np.float32); Ih = np.empty((10000, 10), np.int64)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
td = timeit(lambda: g.search(xq, 32, 10, D=D, I=I))
th = timeit(lambda: g.search(xh, 32, 10, D=Dh, I=Ih))
t0 = time.perf_counter()
for _ in range(20): xq.copy_(torch.from_numpy(xh))
torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / 20 * 1e3
print("device-resident step %.3f ms; host-buffer step %.3f ms (x%.3f); plain pageable H2D of the queries alone %.3f ms; results equal: %s" % (
    td, th, th / td, tc, bool(np.array_equal(Dh, D.cpu().numpy()) and np.array_equal(Ih, I.cpu().numpy()))))
# page-locked buffers (GpuResources::getPinnedMemory's kind)
xp = torch.from_numpy(xh).pin_memory(); Dp = torch.empty((10000, 10), dtype=torch.float32).pin_memory(); Ip = torch.empty((10000, 10), dtype=torch.int64).pin_memory()
xpn, Dpn, Ipn = xp.numpy(), Dp.numpy(), Ip.numpy()
tp = timeit(lambda: g.search(xpn, 32, 10, D=Dpn, I=Ipn))
tpd = timeit(lambda: g.search(xpn, 32, 10, D=D, I=I))          # page-locked queries, device results
tdp = timeit(lambda: g.search(xq, 32, 10, D=Dpn, I=Ipn))        # device queries, page-locked results
def cp_in(): xq.copy_(xp, non_blocking=True)
def cp_out(): Dp.copy_(D, non_blocking=True); Ip.copy_(I, non_blocking=True)
print("page-locked x/D/I %.3f ms (x%.3f); page-locked x only %.3f; page-locked D/I only %.3f; H2D alone %.3f ms; D2H alone %.3f ms; equal: %s" % (
    tp, tp / td, tpd, tdp, timeit(cp_in), timeit(cp_out),
    bool(np.array_equal(Dpn, D.cpu().numpy()) and np.array_equal(Ipn, I.cpu().numpy()))))
