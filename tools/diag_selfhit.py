#!/usr/bin/env python3
"""Diagnostic: do database vectors find themselves?  env: DIM NLIST NB BATCH NPROBE K"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vector_line_quantization_amd as vlq
E = lambda k, v: int(os.environ.get(k, v))
d, nlist, nb, step, nprobe, k = E("DIM", 96), E("NLIST", 131072), E("NB", 4000000), E("BATCH", 4000000), E("NPROBE", 128), E("K", 100)
M, nq = 16, 10000
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(5)
centres = torch.rand((nlist, d), generator=gen, device=dev)
g = vlq.GpuIVFPQ(d, nlist, M, 8)
g.set_stream(torch.cuda.current_stream().cuda_stream)   # the library must run in order with torch's generators
g.set_coarse_centroids(centres)
g.set_pq_centroids(((torch.rand((M, 256, d // M), generator=gen, device=dev) - 0.5) * 0.1).contiguous())
xq = None
for i in range(0, nb, step):
    gb = torch.Generator(device=dev); gb.manual_seed(1000 + i // step)
    n = min(step, nb - i)
    pick = torch.randint(0, nlist, (n,), generator=gb, device=dev)
    xb = (centres[pick] + 0.02 * torch.randn((n, d), generator=gb, device=dev)).contiguous()
    if xq is None: xq, pick0 = xb[:nq].clone(), pick[:nq].clone()
    g.add(xb)
torch.cuda.synchronize()
# 1. coarse assignment of the queries against the generating centre
cd1, k1 = g.coarse_search(xq, 1)
k1 = torch.as_tensor(k1).reshape(-1).cpu().numpy()
print("coarse_search(nprobe=1) == generating centre: %.4f" % (k1 == pick0.cpu().numpy()).mean())
cdp, kp = g.coarse_search(xq, nprobe)
kp = torch.as_tensor(kp).cpu().numpy()
print("generating centre is first of %d probes: %.4f, anywhere in the probes: %.4f" % (
    nprobe, (kp[:, 0] == pick0.cpu().numpy()).mean(), (kp == pick0.cpu().numpy()[:, None]).any(axis=1).mean()))
# 2. is vector i stored in the list of its centre?
p0 = pick0.cpu().numpy()
found = 0
for i in range(200):
    codes, ids = g.get_list(int(p0[i]))
    found += int(i in set(ids.tolist()))
print("vector i stored in the list of its generating centre: %d / 200" % found)
for kk, npb in ((k, nprobe), (10, 32), (1, 1)):
    D, I = g.search(xq, npb, kk)
    I = torch.as_tensor(I).cpu().numpy()
    print("search nprobe=%d k=%d: self-hit@1 %.4f, self anywhere %.4f" % (npb, kk, (I[:, 0] == np.arange(nq)).mean(), (I == np.arange(nq)[:, None]).any(axis=1).mean()))
miss = np.flatnonzero(I[:, 0] != np.arange(nq))[:10]
print("first misses:", miss.tolist(), "their lists:", p0[miss].tolist(), "list lengths:", [g.list_length(int(l)) for l in p0[miss]])
