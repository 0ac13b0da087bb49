#!/usr/bin/env python3
"""How many probe lists do queries adjacent in nearest-centroid order share? (bench data)"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
args = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=float(os.environ.get('SIGMA', 0.03)), gmm_centres=2000,
                             rank=int(os.environ.get('RANK_', 0)), spread=float(os.environ.get('SPREAD', 0.0)))
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(args, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq = bench.gmm(torch, gen, centres, 10000, args.sigma, dev, args.rank, args.spread)
cd, keys = g.coarse_search(xq, 32)
keys = keys.cpu().numpy()
order = np.argsort(keys[:, 0], kind="stable")
ks = keys[order]
for name, kk in (("sorted by top-1", ks), ("unsorted", keys)):
    shared = [len(set(kk[i]) & set(kk[i + 1])) for i in range(0, len(kk) - 1, 2)]
    print(name, "mean shared probes per pair: %.2f of 32; pairs with same top-1: %.3f" % (np.mean(shared), np.mean(kk[0:-1:2, 0] == kk[1::2, 0])))
# lexicographic on (top1, top2)
order2 = np.lexsort((keys[:, 1], keys[:, 0])); k2 = keys[order2]
shared = [len(set(k2[i]) & set(k2[i + 1])) for i in range(0, len(k2) - 1, 2)]
print("sorted by (top1,top2): %.2f" % np.mean(shared))
u, c = np.unique(keys[:, 0], return_counts=True)
print("distinct top-1 cells:", len(u), "queries in cells with >=2 queries: %.3f" % (c[c >= 2].sum() / len(keys)))
os.makedirs("gpurun_out", exist_ok=True)
tag = os.environ.get("TAG", "bench")
np.save("gpurun_out/%s_keys.npy" % tag, keys.astype(np.int32))
np.save("gpurun_out/%s_coarse.npy" % tag, coarse.cpu().numpy())
off = g.list_offsets() if hasattr(g, "list_offsets") else None
if off is None:
    lens = np.array([g.list_length(i) for i in range(args.nlist)], dtype=np.int64)
else:
    lens = np.diff(off)
np.save("gpurun_out/%s_lens.npy" % tag, lens)
