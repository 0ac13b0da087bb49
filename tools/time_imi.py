#!/usr/bin/env python3
"""Inverted-multi-index configuration (BASELINE configs[2] with NBITS=14 NB=1000000000): IMI 2 x NBITS,
table type 2 -- build on the device, time the search, and VERIFY it: the first queries are stored
vectors (self-hit) and a sample of queries is checked bit for bit against the oracle on the lists it
probes, fetched back from the device (tests/scale_checks.py).
   python tools/time_imi.py [nq]   env: NBITS (default 10), NB, BATCH, NPROBE, K, CHECK (sample size, default 16)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vector_line_quantization_amd as vlq
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
E = lambda k, v: int(os.environ.get(k, v))
nbits, nb, nprobe, k, d, M = E("NBITS", 10), E("NB", 4000000), E("NPROBE", 64), E("K", 10), E("DIM", 128), E("M", 16)
nlist = 1 << (2 * nbits)
rng = np.random.default_rng(0)
g = vlq.GpuIVFPQ(d, nlist, M, 8)
g.set_stream(torch.cuda.current_stream().cuda_stream)   # the library must run in order with torch's generators
imi = rng.random((2, 1 << nbits, d // 2), dtype=np.float32)
g.set_imi_centroids(nbits, imi)
pq = ((rng.random((M, 256, d // M), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
g.set_pq_centroids(pq)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
t0 = time.time()
step = E("BATCH", 1000000)
first = None
for i in range(0, nb, step):
    n = min(step, nb - i)
    xbatch = torch.rand((n, d), device="cuda", generator=gen)
    if first is None: first = xbatch[:nq].clone()        # stored vectors 0 .. nq-1 (sequential ids)
    g.add(xbatch)
    if (i // step) % 10 == 9:
        torch.cuda.synchronize(); print("  added %d M in %.1f s" % ((i + n) // 1000000, time.time() - t0), flush=True)
torch.cuda.synchronize()
print("added %d vectors into %d lists in %.1f s, device memory in use %.1f GB" % (nb, nlist, time.time() - t0, (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)
xq = torch.rand((nq, d), device="cuda", generator=gen)
nself = min(nq, first.shape[0]) // 2
xq[:nself] = first[:nself]                              # half the batch: stored vectors, must find themselves
D = torch.empty((nq, k), dtype=torch.float32, device="cuda"); I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
for _ in range(2): g.search(xq, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
g.stats(reset=True); g.profile(not os.environ.get("NOPROF")); g.profile_read(reset=True)      # NOPROF=1: no stage events (the two halves of the coarse stage then run on two streams)
t0 = time.time(); reps = 20 if os.environ.get("NOPROF") else 5
for _ in range(reps): g.search(xq, nprobe, k, D=D, I=I)
torch.cuda.synchronize()
dt = (time.time() - t0) / reps
p = g.profile_read(); _, ncode = g.stats()
print("search %.3f ms per %d queries = %.0f QPS; stages ms: coarse %.3f tables %.3f scan %.3f; ncode/query %.0f" % (
    dt * 1e3, nq, nq / dt, p["coarse_ms"] / reps, p["tables_ms"] / reps, p["scan_ms"] / reps, ncode / reps / nq))

# ---- verification (outside every timed region) ----
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import scale_checks
Ih = I.cpu().numpy()
s1, sk = scale_checks.self_hit(Ih[:nself])
print("self-hit: %d stored vectors as queries: first %.4f, in top-%d %.4f" % (nself, s1, k, sk), flush=True)
ns = E("CHECK", 16)
pick = np.r_[0:ns // 2, nself:nself + ns - ns // 2]
res = scale_checks.check_ivfpq_sample(g, xq[pick].cpu().numpy(), nprobe, k, pq, imi=imi, imi_nbits=nbits)
print("oracle sample check:", res, flush=True)
assert s1 >= 0.99 and res["ok"], "verification failed"
print("VERIFIED")
print("coarse screen (enabled, rows screened, rows done exactly in full):", g.coarse_screen_state())
