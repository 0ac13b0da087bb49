#!/usr/bin/env python3
"""Headline benchmark: queries/sec of IVFPQ search on MI355X.

Workload (BASELINE.json configs[1]): SIFT1M-shaped synthetic data (or the real
.fvecs files when present, --fvecs-dir), d=128, nlist=4096, M=16 x 8 bit,
nprobe=32, k=10, ONE batch of 10 000 queries.  One "step" = one search() of that
batch with queries and results resident in HBM.  N > 1: index replicated on every
GPU, the batch split ceil(nq/N) per rank (IndexProxy.cpp:139-149), one RCCL
all-gather of the per-rank top-k rows per step = strong scaling (the north star's
mode); the weak-scaling figure (10 000 queries per GPU) is measured in the same run
and reported as "other_scaling".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
`python bench.py --gpus N` without a launcher starts the N ranks itself (fresh child
processes, before this process touches the GPU).  A rank count / device / gather
mismatch exits non-zero.

Prints ONE JSON line (rank 0).  torch is used for device memory, streams,
synthetic data/training set-up and torch.distributed; the timed region is the
HIP library behind include/vlq_ivfpq.h only.  The oracle (oracle/) is used for
the cpu_baseline leg and a parity spot check, never in the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level table)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ----------------------------------------------------------------------------
# synthetic SIFT1M-shaped data (generator G1 of SURVEY.md §8d) and index set-up
# ----------------------------------------------------------------------------
_SUBSPACE = {}


def gmm(torch, gen, centres, n, sigma, dev, rank=0, spread=0.0):
    pick = torch.randint(0, centres.shape[0], (n,), generator=gen, device=dev)
    x = centres[pick] + sigma * torch.randn((n, centres.shape[1]), generator=gen, device=dev)
    if rank > 0:
        # low intrinsic dimension (what makes real descriptors rankable by 16-byte codes): most of a
        # point's offset from its centre lies in one fixed `rank`-dimensional subspace
        key = (rank, centres.shape[1], str(dev))
        if key not in _SUBSPACE:
            g2 = torch.Generator(device=dev)
            g2.manual_seed(4242)
            _SUBSPACE[key] = torch.randn((rank, centres.shape[1]), generator=g2, device=dev) / (rank ** 0.5)
        x = x + spread * torch.randn((n, rank), generator=gen, device=dev) @ _SUBSPACE[key]
    return torch.clamp(torch.round(x * 255.0), 0, 255).float()


def kmeans(torch, x, k, niter, gen):
    """Plain Lloyd (set-up only; training is outside the hot path)."""
    n = x.shape[0]
    cent = x[torch.randperm(n, generator=gen, device=x.device)[:k]].clone()
    for _ in range(niter):
        d2 = (cent * cent).sum(1)[None, :] - 2.0 * x @ cent.T
        a = d2.argmin(1)
        cnt = torch.bincount(a, minlength=k).float()
        s = torch.zeros_like(cent).index_add_(0, a, x)
        nz = cnt > 0
        cent[nz] = s[nz] / cnt[nz, None]
        if (~nz).any():   # re-seed empty clusters from random points
            idx = torch.randint(0, n, (int((~nz).sum()),), generator=gen, device=x.device)
            cent[~nz] = x[idx]
    return cent


def build_index(args, dev, xt=None, xb=None, bcast=None):
    """xt / xb given: real vectors (.fvecs); otherwise generator G1 of SURVEY.md section 8(d).
    bcast(t): N > 1 -- the quantizers are trained on rank 0 and broadcast, so that every rank holds the SAME
    index (the k-means below accumulates with float atomics: two GPUs would not train bit-equal centroids)."""
    import torch
    import vector_line_quantization_amd as vlq
    d, nlist, M, nbits = args.d, args.nlist, args.M, 8
    ksub, dsub = 1 << nbits, d // M
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    centres = torch.rand((args.gmm_centres, d), generator=gen, device=dev)
    if xt is None:
        gen.manual_seed(11)
        xt = gmm(torch, gen, centres, args.nt, args.sigma, dev, args.rank, args.spread)
        gen.manual_seed(22)
        xb = gmm(torch, gen, centres, args.nb, args.sigma, dev, args.rank, args.spread)
    t0 = time.time()
    if bcast is None or int(os.environ.get("RANK", "0")) == 0:
        gen.manual_seed(1234)
        coarse = kmeans(torch, xt, nlist, 10, gen)
        # residual PQ training set (IndexIVFPQ.cpp:73-104): subsample to 256*ksub points
        ntr = min(xt.shape[0], 256 * ksub)
        xs = xt[torch.randperm(xt.shape[0], generator=gen, device=dev)[:ntr]]
        a = ((coarse * coarse).sum(1)[None, :] - 2.0 * xs @ coarse.T).argmin(1)
        res = xs - coarse[a]
        pq = torch.stack([kmeans(torch, res[:, m * dsub:(m + 1) * dsub].contiguous(), ksub, 25, gen)
                          for m in range(M)])
    else:
        coarse = torch.empty((nlist, d), dtype=torch.float32, device=dev)
        pq = torch.empty((M, ksub, dsub), dtype=torch.float32, device=dev)
    if bcast is not None:
        coarse, pq = bcast(coarse.contiguous()), bcast(pq.contiguous())
    torch.cuda.synchronize()
    log("trained coarse+PQ in %.1fs" % (time.time() - t0))

    g = vlq.GpuIVFPQ(d, nlist, M, nbits, device=dev.index or 0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.set_coarse_centroids(coarse.contiguous())
    g.set_pq_centroids(pq.contiguous())
    t0 = time.time()
    for i0 in range(0, xb.shape[0], 262144):          # device-side encode + append
        g.add(xb[i0:i0 + 262144].contiguous())
    torch.cuda.synchronize()
    log("added %d vectors in %.1fs (HIP encode path)" % (xb.shape[0], time.time() - t0))
    return g, centres, coarse, pq, xb


def list_stats(g, nlist):
    lens = np.array([g.list_length(i) for i in range(nlist)], dtype=np.float64)
    tot = lens.sum()
    imb = float((lens * lens).sum() * nlist / (tot * tot)) if tot > 0 else 0.0   # IndexIVF.cpp:140-147
    return lens, imb


def oracle_copy(g, args, coarse, pq):
    from oracle import pyoracle
    codes, ids, off = [], [], [0]
    for i in range(args.nlist):
        c, ii = g.get_list(i)
        codes.append(c)
        ids.append(ii)
        off.append(off[-1] + len(ii))
    return pyoracle.OracleIndex(args.d, args.nlist, args.M, 8, coarse.cpu().numpy(), pq.cpu().numpy(),
                                codes=np.concatenate(codes), ids=np.concatenate(ids),
                                list_offsets=np.array(off, np.int64))


def recall(torch, xq, xb, I_h, nb, dev, nr=1000):
    """recall@1 and 1-recall@10 of the first nr queries against exact L2 ground truth"""
    nr = min(nr, xq.shape[0])
    xqr = xq[:nr]
    best = torch.full((nr,), float("inf"), device=dev)
    arg = torch.zeros((nr,), dtype=torch.int64, device=dev)
    qn = (xqr * xqr).sum(1)
    for i0 in range(0, nb, 131072):
        xbb = xb[i0:i0 + 131072]
        d2 = qn[:, None] + (xbb * xbb).sum(1)[None, :] - 2.0 * xqr @ xbb.T
        m, a = d2.min(1)
        upd = m < best
        best[upd] = m[upd]
        arg[upd] = a[upd] + i0
    gt = arg.cpu().numpy()
    return float((I_h[:nr, 0] == gt).mean()), float((I_h[:nr] == gt[:, None]).any(1).mean())


G1_FLAGS = (0.03, 0, 0.0)      # (sigma, rank, spread) of rounds 1-3's headline data: isotropic, recall@1 0.175


def dataset_leg(torch, args, dev, flags, steps=40, with_fp16=False):
    """The same configuration on another setting of the generator, reported beside the headline, never instead of it:
    `first_dataset` = generator G1 as rounds 1-3 ran it (sigma 0.03, isotropic: SURVEY.md 8(d) calls it too noisy --
    recall@1 0.175 -- kept for continuity with BENCH_r01..r03); with_fp16: the opt-in float16 look-up tables on the
    given data."""
    import copy
    a2 = copy.copy(args)
    a2.sigma, a2.rank, a2.spread = flags
    g, centres, coarse, pq, xb = build_index(a2, dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(33)
    xq = gmm(torch, gen, centres, a2.nq, a2.sigma, dev, a2.rank, a2.spread)
    D = torch.empty((a2.nq, a2.k), dtype=torch.float32, device=dev)
    I = torch.empty((a2.nq, a2.k), dtype=torch.int64, device=dev)
    for _ in range(20):
        g.search(xq, a2.nprobe, a2.k, D=D, I=I)
    torch.cuda.synchronize()
    g.stats(reset=True)
    g.profile(True)
    g.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        g.search(xq, a2.nprobe, a2.k, D=D, I=I)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = g.profile_read(reset=True)
    g.profile(False)
    _n, ncode = g.stats(reset=True)
    scan_ms = prof["scan_ms"] / max(1, prof["scan_calls"])
    ncl = ncode / max(1, prof["scan_calls"])
    I32 = I.cpu().numpy()
    r1, r10 = recall(torch, xq, xb, I32, a2.nb, dev)
    out = {"data": "synthetic, generator flags --sigma %g --rank %d --spread %g" % flags, "value": a2.nq / dt,
           "unit": "queries/s", "ms_per_step": dt * 1e3, "scan_kernel_ms": scan_ms, "ncode_per_query": ncl / a2.nq,
           "roofline_frac": (ncl * a2.M / (scan_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS if scan_ms > 0 else 0.0,
           "recall_at_1": r1, "recall_1_at_10": r10}
    if not with_fp16:
        return out
    # The opt-in float16 look-up tables (GpuIndexIVFPQConfig::useFloat16LookupTables) on the same data scaled by
    # 1/256 -- a power of two, so every fp32 product, sum, list assignment and code is the scaled original and the
    # vectors fit the half range (byte-valued coordinates do not: term 2 reaches 1e5).  8 KB rows instead of 16 KB.
    try:
        import vector_line_quantization_amd as vlq
        sc = 1.0 / 256.0
        g2 = vlq.GpuIVFPQ(a2.d, a2.nlist, a2.M, 8, device=dev.index or 0)
        g2.set_stream(torch.cuda.current_stream().cuda_stream)
        g2.set_coarse_centroids((coarse * sc).contiguous())
        g2.set_pq_centroids((pq * sc).contiguous())
        for i in range(0, a2.nb, 250000):
            g2.add((xb[i:i + 250000] * sc).contiguous())
        xq2 = (xq * sc).contiguous()
        g2.set_float16_tables(True)
        for _ in range(20):
            g2.search(xq2, a2.nprobe, a2.k, D=D, I=I)
        torch.cuda.synchronize()
        g2.stats(reset=True)
        g2.profile(True)
        g2.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            g2.search(xq2, a2.nprobe, a2.k, D=D, I=I)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / steps
        prof2 = g2.profile_read(reset=True)
        g2.profile(False)
        I16 = I.cpu().numpy()
        r1h, r10h = recall(torch, xq, xb, I16, a2.nb, dev)
        out["float16_tables"] = {"value": a2.nq / dt2, "unit": "queries/s", "ms_per_step": dt2 * 1e3,
                                 "scan_kernel_ms": prof2["scan_ms"] / max(1, prof2["scan_calls"]),
                                 "recall_at_1": r1h, "recall_1_at_10": r10h,
                                 "labels_equal_fp32_frac": float((I16 == I32).mean()),
                                 "note": "opt-in (vlq_ivfpq_set_float16_tables), data scaled by 1/256 to fit the half range; "
                                         "not the parity build"}
    except Exception as e:     # noqa: BLE001
        out["float16_tables"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return out


def fvecs_read(path):
    """.fvecs / .ivecs reader (tests/demo_sift1M.cpp:40-70): every vector is an int32 dimension
    followed by d 4-byte components.  Returns an [n, d] view (float32; reinterpret for .ivecs)."""
    raw = np.fromfile(path, dtype=np.float32)
    d = int(raw[:1].view(np.int32)[0])
    assert 0 < d < 1000000 and raw.size % (d + 1) == 0, "weird .fvecs file %s" % path
    return np.ascontiguousarray(raw.reshape(-1, d + 1)[:, 1:])


def find_fvecs_dir(arg):
    """The reference driver's data set (tests/demo_sift1M.cpp:112-161: learn / base / query .fvecs +
    groundtruth.ivecs), used instead of the generator when the files are there."""
    for cand in (arg, os.environ.get("SIFT1M_DIR"), "/home/data/sift1m", os.path.join(ROOT, "data", "sift1m")):
        if cand and all(os.path.exists(os.path.join(cand, f)) for f in ("learn.fvecs", "base.fvecs", "query.fvecs")):
            return cand
    return None


class StubIndex:
    """CPU stand-in for the HIP index, ONLY for the launcher / sharding tests (BENCH_STUB=1, gloo):
    deterministic rows that depend on the query alone, so slicing + gathering can be checked against a
    full-batch call.  Never measured, never reported as a result (the JSON line says "stub": true)."""

    def search(self, x, nprobe, k, D=None, I=None):
        import torch
        s = x.double().sum(1, keepdim=True)
        if os.environ.get("BENCH_STUB_CORRUPT") == "1" and int(os.environ.get("RANK", "0")) == 1 and x.shape[0] < 1000:
            s = s + 1.0          # test hook: rank 1 returns wrong rows for its slice -> the gather assertion must fire
        D.copy_((s + torch.arange(k, dtype=torch.float64)[None, :]).float())
        I.copy_((s * 1000.0).long() + torch.arange(k, dtype=torch.int64)[None, :])
        return D, I

    def stats(self, reset=False):
        return 0, 0

    def profile(self, enable=True):
        pass

    def profile_read(self, reset=True):
        return {"coarse_ms": 0.0, "tables_ms": 0.0, "scan_ms": 0.0, "scan_calls": 0}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) and pass rank
    0's JSON line through.  Runs BEFORE this process touches HIP / torch.cuda, and the children are
    new processes (subprocess), never an exec of a process that initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd[1:8])), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def sources_sha():
    import hashlib
    hh = hashlib.sha256()
    for f in ("scan16.hip", "scan16_common.cuh", "wave_topk.cuh", "scan_common.cuh", "walk_order.cuh"):
        with open(os.path.join(ROOT, "vector_line_quantization_amd", "csrc", f), "rb") as fh:
            hh.update(fh.read())
    return hh.hexdigest()


def vlq_traffic():
    """HBM bytes per launch of the VLQ scan from the filed counter passes -- quoted only while the kernel sources they were
    measured on are unchanged (same rule as the headline's roofline.traffic)."""
    import hashlib
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_vlq_traffic.json")), reverse=True):      # newest round first
        try:
            with open(path) as fh:
                rec = json.load(fh)
            hh = hashlib.sha256()
            for f in rec["sources"]:
                with open(os.path.join(ROOT, "vector_line_quantization_amd", "csrc", f), "rb") as fh:
                    hh.update(fh.read())
            if hh.hexdigest() != rec["sources_sha"]:
                continue
            return {name: rec[name]["bytes"] for name in ("fp32_tables", "float16_tables")}
        except (OSError, KeyError, ValueError):
            continue
    return {}


def vlq_leg(torch, dev, nb=16000000, nq=2000, reps=5, nsample=64):
    """The fork's VLQ index at the reference driver's geometry (65 536 centroids x 64 edges =
    4.19 M lines, nLambda 256, M = 16 x 8 bit, nprobe 64, w1 1024, k 128; gpu/test/deep1b16_query.cpp:
    206-208,326,337-340) with a reduced database: build on the device, search with fp32 and with float16
    look-up tables (the drivers' setting), and check a query sample bit for bit against the VLQ oracle on
    the lines the device selected.  Reported beside the headline, never instead of it."""
    import vector_line_quantization_amd as vlq
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scale_checks
    d, nlist, nedge, M, nprobe, w1, k = 96, 65536, 64, 16, 64, 1024, 128
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    cent = torch.rand((nlist, d), generator=gen, device=dev)
    g = vlq.GpuVLQ(d, nlist, M, 8, nedge, 256, device=dev.index or 0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.set_coarse_centroids(cent)
    t0 = time.time()
    ei, ed = g.build_graph()
    lam = np.linspace(-0.2, 1.2, 256).astype(np.float32)
    g.set_lambda_codebook(lam)
    pq = ((torch.rand((M, 256, d // M), generator=gen, device=dev) - 0.5) * 0.2).contiguous()
    g.set_pq_centroids(pq)
    first = None
    for i in range(0, nb, 1000000):
        n = min(1000000, nb - i)
        pick = torch.randint(0, nlist, (n,), device=dev, generator=gen)
        x = (cent[pick] + 0.08 * torch.randn((n, d), device=dev, generator=gen)).contiguous()
        if first is None:
            first = x[:nq].clone()
        g.add(x)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    pick = torch.randint(0, nlist, (nq,), device=dev, generator=gen)
    xq = (cent[pick] + 0.08 * torch.randn((nq, d), device=dev, generator=gen)).contiguous()
    xq[:nq // 2] = first[:nq // 2]                      # half the batch: stored vectors
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    # queries checked bit for bit against the oracle in every leg: half stored vectors, half fresh ones
    sample = np.r_[0:nsample // 2, nq // 2:nq // 2 + nsample - nsample // 2]
    cent_h, pq_h = cent.cpu().numpy(), pq.cpu().numpy()
    out = {"workload": "VLQ, SURVEY C5 geometry: d=96, 65536 centroids x 64 edges, nLambda=256, M=16x8bit, nprobe=64, w1=1024, "
                       "k=128, %d queries per batch, %d synthetic vectors (reduced from 1 B; full size: profiles/)" % (nq, nb),
           "build_s": build_s}
    for name, fp16 in (("fp32_tables", False), ("float16_tables", True)):
        g.set_float16_tables(fp16)
        for _ in range(2):
            g.search(xq, nprobe, w1, k, D=D, I=I)
        torch.cuda.synchronize()
        g.stats(reset=True)
        t1 = time.perf_counter()
        for _ in range(reps):
            g.search(xq, nprobe, w1, k, D=D, I=I)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        ncode = g.stats(reset=True) / reps
        Ih = I.cpu().numpy()
        s1, sk = scale_checks.self_hit(Ih[:nq // 2])
        chk = scale_checks.check_vlq_sample(g, xq[sample].cpu().numpy(), nprobe, w1, k, cent_h, pq_h, lam, ei, ed, fp16=fp16)
        out[name] = {"value": nq / dt, "unit": "queries/s", "ms_per_batch": dt * 1e3, "ncode_per_query": ncode / nq,
                     "self_hit_in_top_k": sk, "oracle_sample_bit_exact": bool(chk["ok"]), "oracle_sample_queries": chk["queries"]}
    # the same geometry at the reference driver's DATABASE size: 238 codes on every one of the 4.19 M lines
    # (998 M codes), loaded as uniformly random (code, lambda) bytes -- the byte traffic of the populated 1 B-vector
    # index without its 150 s device-side build (recall is meaningless here; the oracle check on the device's own
    # lines is not).  This is the leg with a roofline: SURVEY.md 8(d) prices the VLQ scan at 17 B per scanned code.
    per = 238
    nbig = per * nlist * nedge
    codes = torch.empty((nbig, M), dtype=torch.uint8, device=dev)
    for i in range(0, nbig, 1 << 26):
        codes[i:i + (1 << 26)] = torch.randint(0, 256, (min(1 << 26, nbig - i), M), dtype=torch.uint8, device=dev, generator=gen)
    lams = torch.randint(0, 256, (nbig,), dtype=torch.uint8, device=dev, generator=gen)
    g.set_lists(codes, lams, torch.arange(nbig, dtype=torch.int64, device=dev),
                torch.arange(nlist * nedge + 1, dtype=torch.int64, device=dev) * per)
    del codes, lams
    peak = 8000.0
    # bytes from the fabric per scan launch, rocprofv3 --pmc FETCH_SIZE x 2 (profiles/r04_pmc_vlq.txt ->
    # profiles/r04_vlq_traffic.json), quoted only for the workload and the kernel sources it was measured on
    pmc_traffic = vlq_traffic()
    for name, fp16 in (("fp32_tables", False), ("float16_tables", True)):
        g.set_float16_tables(fp16)
        for _ in range(2):
            g.search(xq, nprobe, w1, k, D=D, I=I)
        torch.cuda.synchronize()
        g.stats(reset=True)
        g.profile(True)
        g.profile_read(reset=True)
        t1 = time.perf_counter()
        for _ in range(reps):
            g.search(xq, nprobe, w1, k, D=D, I=I)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        scan_ms, launches = g.profile_read(reset=True)
        g.profile(False)
        kernel_ms = scan_ms / max(1, launches)
        ncode = g.stats(reset=True) / reps
        chk = scale_checks.check_vlq_sample(g, xq[sample].cpu().numpy(), nprobe, w1, k, cent_h, pq_h, lam, ei, ed, fp16=fp16)
        achieved = ncode * (M + 1) / (kernel_ms * 1e-3) / 1e9
        traffic = pmc_traffic.get(name) if nq == 2000 else None
        out["codes_1b_" + name] = {
            "value": nq / dt, "unit": "queries/s", "ms_per_batch": dt * 1e3, "ncode_per_query": ncode / nq,
            "database": "%d synthetic 17-byte codes, %d per line" % (nbig, per),
            "oracle_sample_bit_exact": bool(chk["ok"]), "oracle_sample_queries": chk["queries"],
            "roofline": {"bound": "hbm", "kernel": "line16c_scan_kernel<2, 8, 2, %s>" % ("true" if fp16 else "false"), "kernel_ms": kernel_ms,
                         "launches": launches, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                         "algorithmic_bytes": ncode * (M + 1), "traffic": traffic,
                         "frac_measured": (traffic / (kernel_ms * 1e-3) / 1e9 / peak) if traffic else None,
                         "note": "achieved = scanned codes x 17 B / scan-kernel time (HIP events on the index's stream); the "
                                 "kernel also reads 4 B per code (the stored query-independent la * sum(term 4), line16c.hip) and one "
                                 "%d KB table row per probed centroid (%d per query; the reference's kernel reads two rows per kept "
                                 "line, %d per query) -- `traffic` (PMC FETCH_SIZE x 2) shows the sum, read at the rate HBM "
                                 "delivers for random ~4 KB segments (5.5-5.8 TB/s, MI355X_MICROARCH.md)"
                                 % (8 if fp16 else 16, nprobe, w1)}}
    return out


def _scan_roofline(kernel, kernel_ms, launches, ncode, code_bytes, note):
    achieved = ncode * code_bytes / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "kernel_ms": kernel_ms, "launches": launches, "achieved": achieved, "peak": 8000.0,
            "unit": "GB/s", "frac": achieved / 8000.0, "algorithmic_bytes": ncode * code_bytes, "traffic": None, "note": note}


def imi_c3_leg(torch, dev, nb=1000000000, nq=10000, reps=5, nsample=64):
    """BASELINE configs[2] / SURVEY C3 geometry (tests/sift1b_imi_pq.cpp's quantizer): inverted multi-index 2 x 14 bits =
    2^28 lists over 128 dimensions, M = 16 x 8 bit, nprobe 64, k 10, one batch of 10 000 queries; the database is built on the
    device (add: multi-index assignment + residual encoding + append, ~30 s) from 1 B uniform synthetic vectors -- the config's full
    size.  Half the batch are stored vectors (they must find themselves); a query sample
    is checked bit for bit against the oracle on the lists it probes, fetched back from the device -- coarse cells included (the
    device replays MinSumK).  Reported beside the headline, never instead of it."""
    import vector_line_quantization_amd as vlq
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scale_checks
    d, nbits, M, nprobe, k = 128, 14, 16, 64, 10
    nlist = 1 << (2 * nbits)
    rng = np.random.default_rng(0)
    g = vlq.GpuIVFPQ(d, nlist, M, 8, device=dev.index or 0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    imi = rng.random((2, 1 << nbits, d // 2), dtype=np.float32)
    g.set_imi_centroids(nbits, imi)
    pq = ((rng.random((M, 256, d // M), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
    g.set_pq_centroids(pq)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    t0 = time.time()
    first = None
    step = 10000000
    for i in range(0, nb, step):
        xb = torch.rand((min(step, nb - i), d), device=dev, generator=gen)
        if first is None:
            first = xb[:nq].clone()
        g.add(xb)
        del xb
    torch.cuda.synchronize()
    build_s = time.time() - t0
    xq = torch.rand((nq, d), device=dev, generator=gen)
    xq[:nq // 2] = first[:nq // 2]
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    for _ in range(3):
        g.search(xq, nprobe, k, D=D, I=I)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(4 * reps):                          # uninstrumented: the two halves of the coarse stage run on two streams
        g.search(xq, nprobe, k, D=D, I=I)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / (4 * reps)
    g.stats(reset=True)
    g.profile(True)
    g.profile_read(reset=True)
    for _ in range(reps):
        g.search(xq, nprobe, k, D=D, I=I)
    torch.cuda.synchronize()
    pr = g.profile_read(reset=True)
    g.profile(False)
    _n, ncode = g.stats(reset=True)
    ncode /= reps
    s1, sk = scale_checks.self_hit(I.cpu().numpy()[:nq // 2])
    sample = np.r_[0:nsample // 2, nq // 2:nq // 2 + nsample - nsample // 2]
    chk = scale_checks.check_ivfpq_sample(g, xq[sample].cpu().numpy(), nprobe, k, pq, imi=imi, imi_nbits=nbits)
    out = {"workload": "IVFPQ over an inverted multi-index, SURVEY C3 geometry: d=128, 2 x 14 bits = 2^28 lists, M=16x8bit, nprobe=64, k=10, "
                       "%d queries per batch, %d synthetic vectors added on the device" % (nq, nb),
           "build_s": build_s, "add_vectors_per_s": nb / build_s,
           "value": nq / dt, "unit": "queries/s", "ms_per_batch": dt * 1e3, "ncode_per_query": ncode / nq,
           "stage_ms": {"coarse": pr["coarse_ms"] / reps, "order": pr["tables_ms"] / reps, "scan": pr["scan_ms"] / reps,
                        "note": "HIP events on the index's stream (both halves of the coarse stage on one stream while they are on)"},
           "self_hit_first": s1, "self_hit_in_top_k": sk,
           "oracle_sample_bit_exact": bool(chk["ok"]), "oracle_sample_queries": chk["queries"],
           "oracle_sample_coarse_cells_equal": bool(chk["coarse_keys_equal"] and chk["coarse_dis_bits_equal"]),
           "roofline": _scan_roofline(
               "scan16_short_kernel<1>", pr["scan_ms"] / max(1, pr["scan_calls"]), pr["scan_calls"], ncode, M,
               "achieved = scanned codes x 16 B / scan-kernel time (HIP events).  What this kernel moves is not codes but the "
               "precomputed table: each visited cell's two 8 KB half rows (table type 2, IndexIVFPQ.cpp:645-686), 17 x the code "
               "bytes at 1 B vectors, read at the fabric's ~6.8 TB/s -- counters and what was tried: profiles/r06_scan16_short_pmc.txt")}
    del g
    torch.cuda.empty_cache()
    return out


def deep1b_c4_leg(torch, dev, per_list=4096, nq=10000, reps=3, nsample=64):
    """BASELINE configs[3] / SURVEY C4 geometry (tests/deep1b_imi_pq.cpp's data shape with the flat quantizer of the config line):
    2^17 lists over 96 dimensions (6-dimensional sub-vectors), M = 16 x 8 bit, nprobe 128, k 100, one batch of 10 000 queries.
    The lists are loaded as uniformly random code bytes, `per_list` per list (the byte traffic of a populated index without its
    device-side build; the full-size 1 B-vector build is profiles/r05_deep1b_shape_verified.txt: 7631 codes per list) -- recall
    is meaningless here, the oracle check of a query sample on the probed lists, fetched back from the device, is not."""
    import vector_line_quantization_amd as vlq
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scale_checks
    d, nlist, M, nprobe, k = 96, 1 << 17, 16, 128, 100
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    cent = torch.rand((nlist, d), generator=gen, device=dev)
    pq = ((torch.rand((M, 256, d // M), generator=gen, device=dev) - 0.5) * 0.1).contiguous()
    g = vlq.GpuIVFPQ(d, nlist, M, 8, device=dev.index or 0)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.set_coarse_centroids(cent)
    g.set_pq_centroids(pq)
    nbig = per_list * nlist
    codes = torch.empty((nbig, M), dtype=torch.uint8, device=dev)
    for i in range(0, nbig, 1 << 26):
        codes[i:i + (1 << 26)] = torch.randint(0, 256, (min(1 << 26, nbig - i), M), dtype=torch.uint8, device=dev, generator=gen)
    g.set_lists(codes, torch.arange(nbig, dtype=torch.int64, device=dev), torch.arange(nlist + 1, dtype=torch.int64, device=dev) * per_list)
    del codes
    pick = torch.randint(0, nlist, (nq,), device=dev, generator=gen)
    xq = (cent[pick] + 0.02 * torch.randn((nq, d), device=dev, generator=gen)).contiguous()
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    for _ in range(2):
        g.search(xq, nprobe, k, D=D, I=I)
    torch.cuda.synchronize()
    g.stats(reset=True)
    g.profile(True)
    g.profile_read(reset=True)
    t1 = time.perf_counter()
    for _ in range(reps):
        g.search(xq, nprobe, k, D=D, I=I)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / reps
    pr = g.profile_read(reset=True)
    g.profile(False)
    _n, ncode = g.stats(reset=True)
    ncode /= reps
    info = g.last_scan_info()
    chk = scale_checks.check_ivfpq_sample(g, xq[:nsample].cpu().numpy(), nprobe, k, pq.cpu().numpy(), coarse=cent.cpu().numpy())
    out = {"workload": "IVFPQ, SURVEY C4 geometry: d=96, nlist=2^17, M=16x8bit (dsub 6), nprobe=128, k=100, %d queries per batch, %d "
                       "synthetic 16-byte codes, %d per list (reduced from 1 B vectors = 7631 per list)" % (nq, nbig, per_list),
           "value": nq / dt, "unit": "queries/s", "ms_per_batch": dt * 1e3, "ncode_per_query": ncode / nq,
           "stage_ms": {"coarse": pr["coarse_ms"] / reps, "tables_and_order": pr["tables_ms"] / reps, "scan": pr["scan_ms"] / reps},
           "scan_info": info,
           "oracle_sample_bit_exact": bool(chk["ok"]), "oracle_sample_queries": chk["queries"],
           "roofline": _scan_roofline(
               info.split(" order=")[0].replace("kernel=", "") if isinstance(info, str) else "scan16_kernel", pr["scan_ms"] / max(1, pr["scan_calls"]),
               pr["scan_calls"], ncode, M,
               "achieved = scanned codes x 16 B / scan-kernel time (HIP events on the index's stream); lists of %d codes: the kernel "
               "streams codes, the 16 KB table row per probed list is 1.6 %% of the bytes" % per_list)}
    del g
    torch.cuda.empty_cache()
    return out


TIMED_PROFILE = 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 100 timed steps after 40 warm-up steps (0.11 s of GPU time).  After any idle period (the set-up's host round
    # trips) the GPU needs ~30 steps = 25 ms to reach its clocks: per-step times 0.89, 0.83, 0.79, 0.76, 0.76 ... for
    # consecutive groups of ten (tools/clock_ramp.py), the scan kernel itself 0.70 -> 0.65 ms.  A 20-step region after 3
    # warm-up steps (the round-1/2 default) measured that ramp, 0.81-0.82 ms per step, not the rate
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--nb", type=int, default=1000000)
    ap.add_argument("--nt", type=int, default=100000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    # Generator defaults (round 4): the setting whose CPU-oracle recall@1 is 0.45 +- 0.05 -- SURVEY.md 8(d): "tune sigma once so
    # CPU oracle recall@1 ~ 0.45 +- 0.05 and freeze" -- i.e. in-cluster spread mostly inside a 12-dimensional subspace, as real
    # descriptors have it: recall@1 0.478, 10 559 codes per query, list imbalance 1.3.  Rounds 1-3 ran sigma 0.03 isotropic
    # (recall 0.175, 22 513 codes per query, neighbouring queries sharing 22 of 32 probes: friendlier to the caches); that
    # setting is still measured, as the labelled `first_dataset` leg.
    ap.add_argument("--sigma", type=float, default=0.005)
    ap.add_argument("--gmm-centres", type=int, default=2000)
    ap.add_argument("--rank", type=int, default=12, help="intrinsic dimension of the in-cluster spread (0: isotropic only)")
    ap.add_argument("--spread", type=float, default=0.4)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = ONE batch of --nq queries split ceil(nq/N) per rank (north star, "
                         "IndexProxy.cpp:139-149); weak = --nq queries per rank.  The other mode is reported too.")
    ap.add_argument("--fvecs-dir", default=None, help="directory with learn/base/query.fvecs (+ groundtruth.ivecs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-dataset", action="store_true", help="skip the first_dataset (G1 as in rounds 1-3) and float16-table legs")
    ap.add_argument("--prewarm-ms", type=float, default=100.0,
                    help="untimed clock pre-warm before the --warmup steps: groups of 16 searches until this much time has passed (at least 64)")
    ap.add_argument("--no-host-buffers", action="store_true")
    ap.add_argument("--no-vlq", action="store_true", help="skip the VLQ (configs[4] / SURVEY C5 geometry) leg")
    ap.add_argument("--no-imi", action="store_true", help="skip the multi-index (configs[2] / SURVEY C3 geometry) leg")
    ap.add_argument("--no-deep1b", action="store_true", help="skip the Deep1B-shape (configs[3] / SURVEY C4 geometry) leg")
    ap.add_argument("--cpu-queries", type=int, default=10000)
    args = ap.parse_args()
    defaults = {k: ap.get_default(k) for k in ("nq", "nb", "nt", "d", "nlist", "M", "nprobe", "k", "sigma",
                                               "gmm_centres", "rank", "spread")}
    default_workload = all(getattr(args, k) == v for k, v in defaults.items())

    stub = os.environ.get("BENCH_STUB") == "1"
    global TIMED_PROFILE
    TIMED_PROFILE = int(os.environ.get("BENCH_TIMED_PROFILE", "3"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))            # nothing below has run: no HIP call in this process

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: LOCAL_RANK %d but only %d devices visible" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # BENCH_FORCE_DIST=1 (tests): take the collective path with a single rank too
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    rccl_ranks = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # what the collective library actually sees: N ranks, one distinct device each
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
        if stub:
            me = ("cpu", os.getpid())
        else:
            pr = torch.cuda.get_device_properties(dev)
            me = (torch.cuda.current_device(), str(getattr(pr, "uuid", "")) or str(getattr(pr, "pci_bus_id", local_rank)))
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, me)
        if len(set(seen)) != len(seen):
            raise SystemExit("bench.py: ranks share a device: %s" % (seen,))
        rccl_ranks = dist.get_world_size()

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def _bcast(t):
        dist.broadcast(t, src=0)
        return t
    bcast = _bcast if (use_dist and not stub) else None
    fdir = None if stub else find_fvecs_dir(args.fvecs_dir)
    gt = None
    if stub:
        g, centres, coarse, pq, xb = StubIndex(), None, None, None, None
        imb = 0.0
    elif fdir:
        xt_h, xb_h, xq_h0 = (fvecs_read(os.path.join(fdir, f)) for f in ("learn.fvecs", "base.fvecs", "query.fvecs"))
        args.d, args.nt, args.nb, args.nq = xb_h.shape[1], xt_h.shape[0], xb_h.shape[0], xq_h0.shape[0]
        gpath = os.path.join(fdir, "groundtruth.ivecs")
        if os.path.exists(gpath):
            gt = fvecs_read(gpath).view(np.int32)[:, 0].astype(np.int64)
        g, centres, coarse, pq, xb = build_index(args, dev, xt=torch.from_numpy(xt_h).to(dev), xb=torch.from_numpy(xb_h).to(dev), bcast=bcast)
        default_workload = False
        lens, imb = list_stats(g, args.nlist)
    else:
        g, centres, coarse, pq, xb = build_index(args, dev, bcast=bcast)
        lens, imb = list_stats(g, args.nlist)

    def queries(seed, n):
        if stub:
            gen = torch.Generator()
            gen.manual_seed(seed)
            return torch.rand((n, args.d), generator=gen)
        if fdir:
            return torch.from_numpy(xq_h0[:n]).to(dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        return gmm(torch, gen, centres, n, args.sigma, dev, args.rank, args.spread)

    from vector_line_quantization_amd.sharded import shard_bounds
    prewarm = [0]
    cold = {}

    def run_mode(mode, steps, warmup, instrument):
        """One timed region in `mode`.  strong: every rank holds the SAME batch of nq queries and searches
        its ceil(nq/world) slice; weak: nq queries of its own per rank.  Per step: search the slice, then
        ONE all-gather of the packed (D | I) rows of the slice -- the per-shard top-k to every rank."""
        if mode == "strong":
            xq_full = queries(33, args.nq)
            lo, hi, per = shard_bounds(args.nq, world, rank)
        else:
            xq_full = queries(33 + rank, args.nq)
            lo, hi, per = 0, args.nq, args.nq
        xs = xq_full[lo:hi].contiguous()
        ns = hi - lo
        k = args.k
        nbytes_d = (per * k * 4 + 15) // 16 * 16
        slot = nbytes_d + per * k * 8
        # two buffer sets: the gather of step i (collective stream) overlaps the search of step i+1; a set is
        # reused only after its gather has been waited for
        pack = [torch.zeros((slot,), dtype=torch.uint8, device=dev) for _ in range(2)]
        Ds = [p[:per * k * 4].view(torch.float32).view(per, k) for p in pack]
        Is = [p[nbytes_d:].view(torch.int64).view(per, k) for p in pack]
        gath = [torch.empty((world * slot,), dtype=torch.uint8, device=dev) for _ in range(2)] if use_dist else None
        pend = [None, None]
        nstep = [0]

        def drain():
            for b in range(2):
                if pend[b] is not None:
                    pend[b].wait()
                    pend[b] = None

        def step():
            b = nstep[0] & 1
            nstep[0] += 1
            if pend[b] is not None:
                pend[b].wait()
                pend[b] = None
            if ns > 0:
                g.search(xs, args.nprobe, k, D=Ds[b][:ns], I=Is[b][:ns])
            if use_dist:   # per-shard top-k -> every rank (north star: RCCL all-gather over xGMI)
                pend[b] = dist.all_gather_into_tensor(gath[b], pack[b], async_op=True)

        if not stub and "first_call_ms" not in cold and ns > 0:
            # What a driver that calls search ONCE gets (untimed here, labelled): the first search of the fresh handle in this
            # fresh process -- it builds the precomputed table (IndexIVFPQ::precompute_table: the reference does that at
            # train / read_index time), grows the workspace, and runs with no measured walk period and cold clocks -- and the
            # second, which only lacks the period and the clocks.
            sync()
            t1 = time.perf_counter()
            g.search(xs, args.nprobe, k, D=Ds[0][:ns], I=Is[0][:ns])
            sync()
            cold["first_call_ms"] = (time.perf_counter() - t1) * 1e3
            t1 = time.perf_counter()
            g.search(xs, args.nprobe, k, D=Ds[0][:ns], I=Is[0][:ns])
            sync()
            cold["second_call_ms"] = (time.perf_counter() - t1) * 1e3
        # Clock pre-warm, independent of --steps / --warmup: after the set-up's idle stretches the GPU takes ~30 searches
        # (25 ms) to reach its clocks (tools/clock_ramp.py); a short timed region after a short warm-up measures that ramp
        # (BENCH_r03: 20 steps after 5: 0.804 ms per step against 0.758 for 100 after 40).  Untimed, and every timed step
        # is still timed.
        tpw = time.perf_counter()
        npw = 0
        while True:
            for _ in range(16):
                step()
            drain()
            sync()
            npw += 16
            more = npw < 64 or (time.perf_counter() - tpw) * 1e3 < args.prewarm_ms
            if use_dist:     # every rank runs the same number of (collective) steps
                tcont = torch.tensor([1 if more else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(tcont, op=dist.ReduceOp.MAX)
                more = int(tcont.item()) == 1
            if not more or npw >= 2048:
                break
        prewarm[0] = npw
        for _ in range(warmup):
            step()
        drain()
        sync()
        g.stats(reset=True)
        if instrument:
            # timed region: HIP events around the scan kernel of every 4th step only -- an event record is a barrier in
            # the queue, and two per step cost the step ~8 % (BENCH_TIMED_PROFILE=2: every step, the round-2 setting)
            g.profile(TIMED_PROFILE)
            g.profile_read(reset=True)
        if use_dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()                      # every gather of the timed steps has completed inside the timed region
        sync()
        if use_dist:
            dist.barrier()
        sync()
        elapsed = time.perf_counter() - t0
        prof = g.profile_read(reset=True) if instrument else None
        _nq_stat, ncode = g.stats(reset=True)
        prof_all = None
        if instrument:
            g.profile(1)     # untimed: five more steps with every stage instrumented, for "stage_ms"
            for _ in range(5):
                step()
            drain()
            sync()
            prof_all = g.profile_read(reset=True)
            g.profile(False)
            g.stats(reset=True)
            if world == 1 and ns > 0 and hasattr(g, "reset_walk_state"):
                # warm clocks, cold walk: a search right after the measured walk times were forgotten (the state of a fresh
                # handle: no clock period for the cyclic list-id walk, csrc/walk_order.cuh), against the searches around it
                tc, tw = [], []
                for _ in range(8):
                    g.reset_walk_state()
                    sync()
                    t1 = time.perf_counter()
                    g.search(xs, args.nprobe, k, D=Ds[0][:ns], I=Is[0][:ns])
                    sync()
                    tc.append(time.perf_counter() - t1)
                    for _ in range(3):
                        g.search(xs, args.nprobe, k, D=Ds[0][:ns], I=Is[0][:ns])
                    sync()
                    t1 = time.perf_counter()
                    g.search(xs, args.nprobe, k, D=Ds[0][:ns], I=Is[0][:ns])
                    sync()
                    tw.append(time.perf_counter() - t1)
                tc.sort(); tw.sort()
                cold["cold_walk_ms"] = tc[len(tc) // 2] * 1e3
                cold["warm_single_call_ms"] = tw[len(tw) // 2] * 1e3
                cold["scan_info"] = g.last_scan_info()
                g.stats(reset=True)
        b = (nstep[0] - 1) & 1
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            # ASSERTED: the gathered rows are what a single index returns for the same queries.  strong: rank
            # 0's own search of the WHOLE batch; weak: every rank's slot is its own result.
            gb = gath[b].view(world, slot)
            Dall = torch.cat([gb[r, :per * k * 4].view(torch.float32).view(per, k) for r in range(world)])
            Iall = torch.cat([gb[r, nbytes_d:].view(torch.int64).view(per, k) for r in range(world)])
            if mode == "strong":
                Dchk = torch.empty((args.nq, k), dtype=torch.float32, device=dev)
                Ichk = torch.empty((args.nq, k), dtype=torch.int64, device=dev)
                g.search(xq_full, args.nprobe, k, D=Dchk, I=Ichk)
                sync()
                # rows are laid out rank-major in slots of `per`: row i of the batch sits at (i // per) * per + i % per = i
                ok = bool(torch.equal(Dall[:args.nq], Dchk) and torch.equal(Iall[:args.nq], Ichk))
            else:
                ok = bool(torch.equal(Dall[rank * per:rank * per + ns], Ds[b][:ns]) and
                          torch.equal(Iall[rank * per:rank * per + ns], Is[b][:ns]))
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                raise SystemExit("bench.py: all-gathered results differ from the single-index results (%s scaling)" % mode)
            gather_ok = True
            g.stats(reset=True)
        else:
            gather_ok = None
        total_q = (args.nq if mode == "strong" else world * args.nq) * steps
        return {"elapsed": elapsed, "qps": total_q / elapsed, "ms_per_step": elapsed / steps * 1e3, "prof": prof,
                "prof_all": prof_all, "ncode": ncode, "gather_ok": gather_ok, "xq": xs, "D": Ds[b][:ns], "I": Is[b][:ns],
                "queries_per_rank": ns}

    main_mode = args.scaling if world > 1 else "strong"
    res = run_mode(main_mode, args.steps, args.warmup, instrument=not stub)
    other = None
    if world > 1:
        other_mode = "weak" if main_mode == "strong" else "strong"
        other = run_mode(other_mode, args.steps, args.warmup, instrument=False)
        other["mode"] = other_mode
    elapsed, prof, prof_all, ncode, gather_ok = res["elapsed"], res["prof"], res["prof_all"], res["ncode"], res["gather_ok"]
    xq, D, I = res["xq"], res["D"], res["I"]

    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    if stub:
        out = {"stub": True, "metric": "launcher/sharding self-test (no measurement)", "value": res["qps"], "unit": "queries/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
               "scaling": main_mode, "rccl_ranks": rccl_ranks, "config": {"all_gather_check": gather_ok,
               "queries_per_rank": res["queries_per_rank"]}}
        if other is not None:
            out["other_scaling"] = {"scaling": other["mode"], "value": other["qps"], "all_gather_check": other["gather_ok"]}
        print(json.dumps(out), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    steps = args.steps
    qps = res["qps"]
    nq_rank = res["queries_per_rank"]
    # scan launches per step from the five fully instrumented steps (the timed region times a sample of its launches)
    launches_per_step = max(1, prof_all["scan_calls"] // 5)
    ncode_per_launch = ncode / max(1, steps * launches_per_step)
    scan_ms = prof["scan_ms"] / max(1, prof["scan_calls"])
    code_bytes = ncode_per_launch * args.M           # B_scan = ncode * code_size
    achieved = code_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    lut_bytes = nq_rank * args.nprobe * args.M * 256 * 4.0
    # HBM traffic of the scan kernel per launch: PMC counters cannot be read from inside this process, so
    # the figure comes from the committed rocprofv3 PMC passes of this same command (profiles/pmc_passes.sh;
    # FETCH_SIZE corrected x2 as the gfx950 guide prescribes) -- and ONLY while it still describes this run:
    # default workload, one GPU, and the scan-kernel sources unchanged since the passes were taken
    traffic, traffic_source = None, "null: no committed PMC passes match this workload and these kernel sources"
    import glob
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_scan_traffic.json")))      # newest round last
    if tfiles and default_workload and world == 1 and not fdir:
        tname = "profiles/" + os.path.basename(tfiles[-1])
        try:
            tj = json.load(open(tfiles[-1]))
            gen_now = [args.sigma, args.rank, args.spread]
            if tj.get("generator", list(G1_FLAGS)) != gen_now:
                traffic_source = "null: %s was measured on another setting of the data generator" % tname
            elif tj.get("sources_sha256") == sources_sha():
                traffic = tj["hbm_bytes_per_launch"]
                at = tj.get("kernel_ms_in_same_refresh_run")
                traffic_source = "%s (rocprofv3 --pmc passes of this command, same kernel sources%s)" % (
                    tname, "; the kernel averaged %.4f ms in that run's kernel trace" % at if at else "")
            else:
                traffic_source = "null: scan-kernel sources changed since %s was taken" % tname
        except Exception:
            traffic = None

    out = {
        "metric": "queries/sec @ recall@1 (SIFT1M, nlist=4096 m=16 nprobe=32 k=10)",
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": args.warmup, "prewarm_steps": prewarm[0],
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": main_mode,
        "vs_baseline": None, "dtype": "f32",
        "data": ("SIFT1M .fvecs from %s" % fdir) if fdir else
                ("synthetic (SIFT1M-shaped GMM bytes, generator flags --sigma %g --rank %d --spread %g; no dataset on the box)"
                 % (args.sigma, args.rank, args.spread)),
        "rccl_ranks": rccl_ranks,
        "config": {"workload": "configs[1]: %s, d=%d nb=%d nlist=%d M=%dx8bit "
                               "nprobe=%d k=%d, %s, precomputed-table mode 1"
                               % ("SIFT1M .fvecs" if fdir else "SIFT1M-shaped GMM bytes", args.d, args.nb, args.nlist,
                                  args.M, args.nprobe, args.k,
                                  ("ONE batch of %d queries split %d per GPU (strong scaling)" % (args.nq, nq_rank))
                                  if main_mode == "strong" else ("%d queries per GPU (weak scaling)" % args.nq)),
                   "parallelism": "index replicated, queries sharded x%d, one RCCL all-gather of the packed top-k rows per step" % world,
                   "list_imbalance": round(imb, 3), "ncode_per_query": ncode_per_launch / max(1, nq_rank)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "vlq::scan16_kernel", "kernel_ms": scan_ms,
                     # `achieved` / `frac` are ALGORITHMIC code bytes over kernel time: an effective rate,
                     # part of it served by L2 / Infinity Cache.  frac_measured = bytes the fabric counters saw over the
                     # same time (null when no counter run matches this build and workload).
                     "frac_measured": (traffic / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (traffic and scan_ms > 0) else None,
                     "algorithmic_bytes": code_bytes,
                     "lut_bytes_separate": lut_bytes,
                     "lut_plus_code_GBps": (code_bytes + lut_bytes) / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0,
                     # the kernel's other resource: one random 32-bit LDS gather per code byte.  Reference rate = the
                     # measured gather rate of the LDS pipe for uniformly random byte indices (tools/micro/lds_gather.hip:
                     # 9.98 lanes per clock and CU at 16 waves per CU, 2.4 GHz nominal, 256 CUs; bank arithmetic: 64 /
                     # (2 x 3.15)); long-list data runs at 1.07 of it, so it is a yardstick, not a ceiling.  It is NOT what
                     # bounds the headline shape: with every gather compiled out the kernel takes 0.512 ms against 0.520
                     # (profiles/r05_scan16_phases.txt section 4: an LDS / issue floor of 0.43 ms and a fabric-traffic
                     # floor of 0.485 ms overlap to 0.52) -- reported beside the HBM fraction the contract asks for
                     "lds_gather": {"achieved_per_clock_cu": code_bytes / (scan_ms * 1e-3) / 256 / 2.4e9 if scan_ms > 0 else 0.0,
                                    "peak_per_clock_cu": 9.98,
                                    "frac": code_bytes / (scan_ms * 1e-3) / 256 / 2.4e9 / 9.98 if scan_ms > 0 else 0.0}},
        "stage_ms": {"coarse": prof_all["coarse_ms"] / 5, "tables": prof_all["tables_ms"] / 5,
                     "scan": prof_all["scan_ms"] / 5, "note": "from 5 extra untimed steps with every stage instrumented"},
    }

    out["config"]["all_gather_check"] = gather_ok       # asserted above: a wrong gather exits non-zero
    # the cold path (outside the timed region): single synchronous calls, host-timed -- each carries ~20 us of launch and
    # synchronisation that the pipelined timed steps do not, hence `warm_single_call_ms` beside `cold_walk_ms`
    for kk in ("first_call_ms", "second_call_ms", "cold_walk_ms", "warm_single_call_ms"):
        if kk in cold:
            out[kk] = cold[kk]
    if "cold_walk_ms" in cold:
        out["cold_walk_over_warm"] = cold["cold_walk_ms"] / cold["warm_single_call_ms"]
        out["config"]["scan_info"] = cold.get("scan_info")
    # continuity: which earlier records this headline can be compared with.  Rounds 1-3 ran the generator setting
    # that is now the `first_dataset` leg; rounds 4+ run this one, and since round 4 an untimed clock pre-warm precedes --warmup.
    gen_id = "fvecs" if fdir else "gmm2000-sigma%g-rank%d-spread%g" % (args.sigma, args.rank, args.spread)
    out["dataset"] = {"id": gen_id, "generator_flags": None if fdir else [args.sigma, args.rank, args.spread], "headline_comparable_with": ["BENCH_r04"] if default_workload and not fdir else [],
                      "first_dataset_leg_continues": ["BENCH_r01", "BENCH_r02", "BENCH_r03"],
                      "comparable_with_rounds_1_to_3": False}
    if other is not None:
        out["other_scaling"] = {"scaling": other["mode"], "value": other["qps"], "unit": "queries/s",
                                "ms_per_step": other["ms_per_step"], "queries_per_gpu_per_step": other["queries_per_rank"],
                                "all_gather_check": other["gather_ok"]}
    # everything below is outside the timed region; a failure there must not lose the measured line
    try:
        # ---- parity spot check + recall + CPU baseline (outside the timed region) ----
        ox = oracle_copy(g, args, coarse, pq)
        xq_h = xq.cpu().numpy()
        D_h, I_h = D.cpu().numpy(), I.cpu().numpy()
        nchk = min(512, xq_h.shape[0])
        Do, Io = ox.search(xq_h[:nchk], args.nprobe, args.k, canonical=True)
        out["parity"] = {"queries_checked": nchk,
                         "distance_bits_equal": bool(np.array_equal(D_h[:nchk].view(np.uint32), Do.view(np.uint32))),
                         "label_mismatches": int((I_h[:nchk] != Io).sum())}
        if gt is not None and world == 1:
            nr = min(I_h.shape[0], gt.shape[0])
            r1, r10 = float((I_h[:nr, 0] == gt[:nr]).mean()), float((I_h[:nr] == gt[:nr, None]).any(1).mean())
        else:
            r1, r10 = recall(torch, xq, xb, I_h, args.nb, dev)
        out["config"]["recall_at_1"] = r1
        out["config"]["recall_1_at_10"] = r10
        # north_star's "recall@1 >= 0.90" needs a re-ranking stage (IndexIVFPQR, IndexIVFPQ.cpp:1289-1479, what
        # demo_sift1M.cpp:98 selects) that is outside SURVEY.md section 8: 16-byte codes alone do not reach it
        out["config"]["recall_target_met"] = bool(r1 >= 0.90)
        # the float16 screen of the coarse stage (speed only; csrc/coarse_screen.hip): in use? rows it decided / handed on
        en, rows_s, und = g.coarse_screen_state()
        out["config"]["coarse_screen"] = {"enabled": en, "rows_screened": rows_s, "rows_done_exactly_in_full": und}
        if world == 1 and not args.no_host_buffers:
            # the reference drivers' calling convention: queries and results in (pageable) HOST memory;
            # PCIe-inclusive, reported beside `value`, never instead of it
            Dh = np.empty((xq_h.shape[0], args.k), np.float32)
            Ih = np.empty((xq_h.shape[0], args.k), np.int64)
            for _ in range(10):
                g.search(xq_h, args.nprobe, args.k, D=Dh, I=Ih)
            t1 = time.perf_counter()
            for _ in range(40):
                g.search(xq_h, args.nprobe, args.k, D=Dh, I=Ih)
            dth = (time.perf_counter() - t1) / 40
            out["value_host_buffers"] = xq_h.shape[0] / dth
            out["host_buffers"] = {"ms_per_step": dth * 1e3, "ratio_to_device_resident": dth * 1e3 / out["ms_per_step"],
                                   "results_equal_device_resident": bool(np.array_equal(Dh, D_h) and np.array_equal(Ih, I_h)),
                                   "note": "same batch, x / D / I in pageable host memory (numpy), synchronous call"}
            # the same call with the three arrays in PAGE-LOCKED host memory (what GpuResources::getPinnedMemory hands
            # a driver, gpu/GpuResources.h:40): the copies are asynchronous, the kernels queue up behind them
            xp = torch.from_numpy(xq_h).pin_memory()
            Dp = torch.empty((xq_h.shape[0], args.k), dtype=torch.float32).pin_memory()
            Ip = torch.empty((xq_h.shape[0], args.k), dtype=torch.int64).pin_memory()
            xpn, Dpn, Ipn = xp.numpy(), Dp.numpy(), Ip.numpy()
            for _ in range(10):
                g.search(xpn, args.nprobe, args.k, D=Dpn, I=Ipn)
            t1 = time.perf_counter()
            for _ in range(40):
                g.search(xpn, args.nprobe, args.k, D=Dpn, I=Ipn)
            dtp = (time.perf_counter() - t1) / 40
            out["host_buffers_pinned"] = {"ms_per_step": dtp * 1e3, "ratio_to_device_resident": dtp * 1e3 / out["ms_per_step"],
                                          "value": xq_h.shape[0] / dtp,
                                          "results_equal_device_resident": bool(np.array_equal(Dpn, D_h) and np.array_equal(Ipn, I_h)),
                                          "note": "same batch, x / D / I in page-locked host memory, synchronous call"}
            g.stats(reset=True)
        if world == 1 and default_workload and not fdir and not args.no_second_dataset:
            out["first_dataset"] = dataset_leg(torch, args, dev, G1_FLAGS)
            out["first_dataset"]["note"] = ("generator G1 as rounds 1-3 ran it (their headline): recall@1 0.175, below SURVEY.md "
                                            "8(d)'s 0.45 +- 0.05; kept for continuity with BENCH_r01..r03")
            try:
                out["float16_tables"] = dataset_leg(torch, args, dev, (args.sigma, args.rank, args.spread), with_fp16=True)["float16_tables"]
            except Exception as e:     # noqa: BLE001
                out["float16_tables"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

        if world == 1 and default_workload and not fdir and not args.no_vlq:
            try:
                out["vlq_c5_geometry"] = vlq_leg(torch, dev)
            except Exception as e:     # noqa: BLE001 -- never lose the headline over the extra leg
                out["vlq_c5_geometry"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if world == 1 and default_workload and not fdir and not args.no_imi:
            try:
                out["imi_c3_geometry"] = imi_c3_leg(torch, dev)
            except Exception as e:     # noqa: BLE001 -- never lose the headline over an extra leg
                out["imi_c3_geometry"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if world == 1 and default_workload and not fdir and not args.no_deep1b:
            try:
                out["deep1b_c4_geometry"] = deep1b_c4_leg(torch, dev)
            except Exception as e:     # noqa: BLE001
                out["deep1b_c4_geometry"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
            from oracle import pyoracle, refbench
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count()
            ncpu = min(args.cpu_queries, xq_h.shape[0])
            cand_threads = sorted({min(cores, c) for c in (8, 16, 32, 64, 128, 256)})
            done = False
            if refbench.available():
                # the REFERENCE's own CPU IndexIVFPQ (oracle/_ref, compiled from the reference's sources by
                # oracle/ref.mk) on this box's host cores: same index (handed over in the reference's file
                # format), same queries, same nprobe / k
                import tempfile
                try:
                    with tempfile.TemporaryDirectory() as td:
                        ipath = os.path.join(td, "bench.faissindex")
                        refbench.write_ivfpq_index(ipath, ox.coarse_centroids,
                                                   pq.cpu().numpy(), 8, ox.codes, ox.ids, ox.list_offsets)
                        best_t, best_dt = cand_threads[0], float("inf")
                        for nt in cand_threads:                       # thread count that serves this host best
                            _d, _i, secs, _m = refbench.run_reference(ipath, xq_h[:min(ncpu, 2000)], args.nprobe, args.k, 1, nt)
                            if secs[0] < best_dt:
                                best_t, best_dt = nt, float(secs[0])
                        Dr, Ir, secs, meta = refbench.run_reference(ipath, xq_h[:ncpu], args.nprobe, args.k, 3, best_t)
                    med = float(np.sort(secs)[1])
                    out["cpu_baseline"] = {"value": ncpu / med, "unit": "queries/s", "cores": int(meta[2]), "kind": "reference",
                                           "sample": "%d of the same queries, same index (reference file format), median of 3 "
                                                     "IndexIVFPQ::search calls of the compiled reference (oracle/_ref, MKL), "
                                                     "%d OpenMP threads = fastest of %s tried; affinity mask %d cpus, host %d "
                                                     "logical cpus; use_precomputed_table=%d, ncode/query=%.0f"
                                                     % (ncpu, int(meta[2]), cand_threads, cores, os.cpu_count(), int(meta[0]),
                                                        float(meta[1]) / ncpu)}
                    # the reference's answers beside the MI355X's, all sampled queries: labels up to exact-distance
                    # ties and the BLAS coarse stage's rounding, distances to 1e-4 relative
                    Dg, Ig = D_h[:ncpu], I_h[:ncpu]
                    same = Ig == Ir
                    rel = np.abs(Dg[same] - Dr[same]) / np.maximum(np.abs(Dr[same]), 1e-20)
                    # labels inside a group of bit-equal distances may come in any order, and a group cut by
                    # the k-th place may keep different members (the reference's heap history decides, SURVEY.md
                    # section 7): compare groups as sets, groups reaching the last place by distance only
                    slots_ok = 0
                    for r in range(ncpu):
                        a = 0
                        while a < args.k:
                            b = a + 1
                            while b < args.k and Dr[r, b] == Dr[r, a]:
                                b += 1
                            if np.array_equal(Dg[r, a:b].view(np.uint32), Dr[r, a:b].view(np.uint32)):
                                if b == args.k:
                                    slots_ok += b - a
                                else:
                                    slots_ok += len(set(Ig[r, a:b].tolist()) & set(Ir[r, a:b].tolist()))
                            a = b
                    out["parity"]["vs_reference_cpu"] = {
                        "queries": int(ncpu), "labels_equal_slot_by_slot": float(same.mean()),
                        "labels_equal_up_to_exact_distance_ties": slots_ok / float(ncpu * args.k),
                        "distance_bits_equal_frac": float((Dg.view(np.uint32) == Dr.view(np.uint32)).mean()),
                        "max_rel_distance_error_on_equal_labels": float(rel.max()) if rel.size else 0.0}
                    done = True
                except Exception as e:     # no MKL on this host, driver failure: fall back to the port, say so
                    out["cpu_baseline_note"] = "reference run failed (%s); port timed instead" % (str(e)[:200],)
            if not done:
                # pick the OpenMP thread count that serves the CPU best on this box (the affinity mask
                # can be wider than the container's cpu share)
                best_t, best_dt = 1, float("inf")
                for nt in cand_threads:
                    pyoracle.set_num_threads(nt)
                    ox.search(xq_h[:min(ncpu, 2000)], args.nprobe, args.k)
                    t1 = time.perf_counter()
                    ox.search(xq_h[:min(ncpu, 2000)], args.nprobe, args.k)
                    dt = time.perf_counter() - t1
                    if dt < best_dt:
                        best_t, best_dt = nt, dt
                pyoracle.set_num_threads(best_t)
                ts = []
                for _ in range(3):
                    t1 = time.perf_counter()
                    ox.search(xq_h[:ncpu], args.nprobe, args.k)
                    ts.append(time.perf_counter() - t1)
                ts.sort()
                out["cpu_baseline"] = {"value": ncpu / ts[1], "unit": "queries/s", "cores": pyoracle.num_threads(),
                                       "kind": "port",
                                       "sample": "%d of the same queries, same index, median of 3 search() calls, "
                                                 "oracle restatement (-O3 -fopenmp, %d OpenMP threads = fastest of "
                                                 "8..%d tried; affinity mask %d cpus, host %d logical cpus)"
                                                 % (ncpu, pyoracle.num_threads(), cores, cores, os.cpu_count())}
    except Exception as exc:   # noqa: BLE001
        import traceback
        out["post_run_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
        traceback.print_exc(file=sys.stderr)
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
