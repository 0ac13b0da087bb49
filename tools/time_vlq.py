#!/usr/bin/env python3
"""VLQ (line quantization) at the reference driver's geometry (NLIST=65536 NEDGE=64
NB=1000000000): build on the device, time the search, and VERIFY it -- a sample of queries is checked
bit for bit against the VLQ oracle on the lines it selects, fetched back from the device, and stored
vectors used as queries must come back (tests/scale_checks.py).
   python tools/time_vlq.py [nq] [reps]   env: NB, NLIST, NEDGE, NPROBE, W1, K, D, CHECK (sample size, default 6),
   FP16=1: float16 look-up tables (GpuIndexIVFPQConfig::useFloat16LookupTables, the reference drivers' setting)
   ROWS=1|2: vlq_line_set_row_mode (default 0 = automatic)
   SYNTH=1: the database is NB uniformly random (code, lambda) bytes loaded with set_lists, NB / lines per line --
   the byte traffic of a populated index without the 150 s device-side build of 1 B vectors (recall is
   meaningless then; the oracle check of a query sample on the device's own lines still applies)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vector_line_quantization_amd as vlq

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
E = lambda k, v: int(os.environ.get(k, v))
d, nlist, nedge, M, nbits = E("D", 96), E("NLIST", 4096), E("NEDGE", 16), E("M", 16), E("NBITS", 8)
nb, nprobe, w1, k = E("NB", 4000000), E("NPROBE", 64), E("W1", 1024), E("K", 128)
rng = np.random.default_rng(0)
g = vlq.GpuVLQ(d, nlist, M, nbits, nedge, 256)
g.set_stream(torch.cuda.current_stream().cuda_stream)   # the library must run in order with torch's generators
cent = rng.random((nlist, d), dtype=np.float32)
g.set_coarse_centroids(cent)
ei, ed = g.build_graph()
lam = np.linspace(-0.2, 1.2, 256).astype(np.float32)
g.set_lambda_codebook(lam)
pq = ((rng.random((M, 1 << nbits, d // M), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
g.set_pq_centroids(pq)
t0 = time.time()
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
synth = bool(E("SYNTH", 0))
if synth:
    nl = nlist * nedge
    per = nb // nl
    nb = per * nl
    codes = torch.empty((nb, M), dtype=torch.uint8, device="cuda")
    for i in range(0, nb, 1 << 26):
        codes[i:i + (1 << 26)] = torch.randint(0, 256, (min(1 << 26, nb - i), M), dtype=torch.uint8, device="cuda", generator=gen)
    lams = torch.randint(0, 256, (nb,), dtype=torch.uint8, device="cuda", generator=gen)
    g.set_lists(codes, lams, torch.arange(nb, dtype=torch.int64, device="cuda"),
                torch.arange(nl + 1, dtype=torch.int64, device="cuda") * per)
    del codes, lams
    first = torch.empty((0, d), device="cuda")
else:
    step = 1000000
    for i in range(0, nb, step):
        n = min(step, nb - i)
        pick = torch.randint(0, nlist, (n,), device="cuda", generator=gen)
        x = torch.from_numpy(cent).cuda()[pick] + 0.08 * torch.randn((n, d), device="cuda", generator=gen)
        if i == 0: first = x[:nq].clone()                   # stored vectors 0 .. nq-1 (sequential ids)
        g.add(x.contiguous())
torch.cuda.synchronize()
print("%s %d vectors in %.1f s (%d lines, %.1f per line)" % ("loaded (synthetic codes)" if synth else "added", nb, time.time() - t0, nlist * nedge, nb / (nlist * nedge)), flush=True)
pick = torch.randint(0, nlist, (nq,), device="cuda", generator=gen)
xq = (torch.from_numpy(cent).cuda()[pick] + 0.08 * torch.randn((nq, d), device="cuda", generator=gen)).contiguous()
nself = min(nq, first.shape[0]) // 2
xq[:nself] = first[:nself]                              # half the batch: stored vectors
if E("PARTS", 0): g.set_scan_parts(E("PARTS", 0))     # workgroups per query of the default scan (0 = automatic)
if E("ROWS", 0): g.set_row_mode(E("ROWS", 0))            # 1: term-2 rows from the stored table, 2: rebuilt in the kernel
fp16 = bool(E("FP16", 0))
if fp16 or (M == 16 and nbits == 8): g.set_float16_tables(fp16)
print("look-up tables: %s" % ("float16" if fp16 else "float32"), flush=True)
D = torch.empty((nq, k), dtype=torch.float32, device="cuda")
I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
for _ in range(2):
    g.search(xq, nprobe, w1, k, D=D, I=I)
torch.cuda.synchronize()
g.stats(reset=True)
g.profile(True)
g.profile_read(reset=True)
t0 = time.time()
for _ in range(reps):
    g.search(xq, nprobe, w1, k, D=D, I=I)
torch.cuda.synchronize()
dt = (time.time() - t0) / reps
scan_ms, launches = g.profile_read(reset=True)
g.profile(False)
print("scan kernel: %.3f ms per launch (%d launches, HIP events on the index's stream)" % (scan_ms / max(launches, 1), launches))
ncode = g.stats() / reps
print("search: %.3f ms per %d queries = %.0f QPS; ncode/query=%.0f -> %.0f GB/s (code + lambda bytes)" % (
    dt * 1e3, nq, nq / dt, ncode / nq, ncode * (M + 1) / dt / 1e9))

# ---- verification (outside every timed region) ----
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import scale_checks
Ih = I.cpu().numpy()
s1, sk = scale_checks.self_hit(Ih[:nself]) if nself else (1.0, 1.0)
print("self-hit: %d stored vectors as queries: first %.4f, in top-%d %.4f" % (nself, s1, k, sk), flush=True)
ns = E("CHECK", 6)
pick = np.r_[0:ns // 2, nself:nself + ns - ns // 2]
res = scale_checks.check_vlq_sample(g, xq[pick].cpu().numpy(), nprobe, w1, k, cent, pq, lam, ei, ed, fp16=fp16)
print("oracle sample check:", res, flush=True)
assert sk >= 0.9 and res["ok"], "verification failed"
print("VERIFIED")
