import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = types.SimpleNamespace(d=128, nlist=4096, M=16, nt=100000, nb=1000000, sigma=0.03, gmm_centres=2000, rank=0, spread=0.0)
dev = torch.device("cuda", 0)
g, centres, coarse, pq, xb = bench.build_index(a, dev)
gen = torch.Generator(device=dev); gen.manual_seed(33)
xq = bench.gmm(torch, gen, centres, 10000, a.sigma, dev)
for k in [int(v) for v in os.environ.get("KS", "256,257,512,1000").split(",")]:
    D = torch.empty((10000, k), dtype=torch.float32, device=dev); I = torch.empty((10000, k), dtype=torch.int64, device=dev)
    g.search(xq, 32, k, D=D, I=I); torch.cuda.synchronize()
    g.profile(True); g.profile_read(reset=True)
    t0=time.time(); g.search(xq, 32, k, D=D, I=I); torch.cuda.synchronize(); dt=time.time()-t0
    p = g.profile_read(); g.profile(False)
    print(k, "total %.2f ms"%(dt*1e3), {kk: round(v,3) for kk,v in p.items()}, flush=True)
