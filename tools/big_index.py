#!/usr/bin/env python3
"""Scale check on one MI355X: build a multi-GB IVFPQ index with device-side add in 1 M batches and
search it (64-bit offsets, slack relayouts, 1 GiB term2).   python tools/big_index.py [nb] [nlist]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vector_line_quantization_amd as vlq
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 50000000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
d, M, nq, nprobe, k = int(os.environ.get('DIM', 128)), 16, 10000, int(os.environ.get('NPROBE', 32)), int(os.environ.get('K', 10))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(5)
centres = torch.rand((nlist, d), generator=gen, device=dev)
g = vlq.GpuIVFPQ(d, nlist, M, 8)
g.set_stream(torch.cuda.current_stream().cuda_stream)   # the library must run in order with torch's generators
t0 = time.time()
g.set_coarse_centroids(centres)
print("set_coarse_centroids (incl. spatial rank of %d lists): %.1f s" % (nlist, time.time() - t0), flush=True)
pq_t = ((torch.rand((M, 256, d // M), generator=gen, device=dev) - 0.5) * 0.1).contiguous()
g.set_pq_centroids(pq_t)
def batch(i, n):
    gb = torch.Generator(device=dev); gb.manual_seed(1000 + i)
    pick = torch.randint(0, nlist, (n,), generator=gb, device=dev)
    return (centres[pick] + 0.02 * torch.randn((n, d), generator=gb, device=dev)).contiguous()
t0 = time.time()
step = int(os.environ.get('BATCH', 1000000))
xq = None
for i in range(0, nb, step):
    xb = batch(i // step, min(step, nb - i))
    if xq is None: xq = xb[:nq].clone()           # the first nq database vectors
    g.add(xb)
    del xb
    if (i // step) % 10 == 9 or i + step >= nb:
        torch.cuda.synchronize(); print("  added %d M in %.1f s" % ((i + step) // 1000000, time.time() - t0), flush=True)
torch.cuda.synchronize()
dt = time.time() - t0
print("added %d vectors in %.1f s (%.2f M vectors/s), ntotal=%d, device memory in use %.1f GB" % (
    nb, dt, nb / dt / 1e6, g.ntotal, (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)
lens = np.array([g.list_length(i) for i in range(0, nlist, 97)])
print("sampled list lengths: mean %.1f max %d" % (lens.mean(), lens.max()))
freed = g.reclaim_memory()
print("reclaim_memory gave back %.2f GB of append slack" % (freed / 1e9), flush=True)
D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
for _ in range(2): g.search(xq, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
g.stats(reset=True)
t0 = time.time()
for _ in range(5): g.search(xq, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
dt = (time.time() - t0) / 5
_n, ncode = g.stats()
Ih = I.cpu().numpy()
self_hit = float((Ih[:, 0] == np.arange(nq)).mean())
self_in = float((Ih == np.arange(nq)[:, None]).any(axis=1).mean())
print("search: %.3f ms per %d queries = %.2f M queries/s, %.0f codes per query, self-hit@1 %.3f, self in top-%d %.3f" % (
    dt * 1e3, nq, nq / dt / 1e6, ncode / 5 / nq, self_hit, k, self_in))

# ---- verification (outside every timed region): a query sample against the oracle on the probed lists ----
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import scale_checks
res = scale_checks.check_ivfpq_sample(g, xq[:int(os.environ.get("CHECK", 8))].cpu().numpy(), nprobe, k, pq_t.cpu().numpy(),
                                      coarse=centres.cpu().numpy())
print("oracle sample check:", res, flush=True)
assert self_hit >= 0.99 and res["ok"], "verification failed"
print("VERIFIED")
