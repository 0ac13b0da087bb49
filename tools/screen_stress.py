#!/usr/bin/env python3
"""Randomised check of the coarse stage's float16 screen against the matrix path of the same library (the switch is speed
only: vlq_ivfpq_set_coarse_screen): random shapes, scales, offsets, cluster structure, quantised coordinates (ties),
duplicated centroids, queries on centroids; keys and distances must be bit-identical.  Also nprobe = 1 (the nearest-
centroid screen behind add / encode).   python tools/screen_stress.py [cases] [seed]   (IMI=1: the two halves of a multi-index instead)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vector_line_quantization_amd as vlq

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
bad = 0
kinds = ("uniform", "gauss", "clusters", "offset", "quantised", "lowrank", "tiny", "huge")
t0 = time.time()
for case in range(ncase):
    d = int(rng.choice([4, 8, 16, 32, 64, 96, 100, 128]))
    nlist = 64 * int(rng.choice([4, 5, 16, 17, 64, 100, 128, 129, 256, 400]))
    nprobe = int(rng.choice([2, 3, 8, 16, 31, 32, 64, 65, 100, 128]))
    nprobe = min(nprobe, nlist // 2)
    nq = int(rng.choice([2048, 2049, 2200, 3000, 4096]))
    kind = kinds[case % len(kinds)]
    if kind == "uniform":
        cent = rng.random((nlist, d)); xq = rng.random((nq, d))
    elif kind == "gauss":
        cent = rng.standard_normal((nlist, d)); xq = rng.standard_normal((nq, d))
    elif kind == "clusters":
        c0 = rng.random((32, d)); cent = c0[rng.integers(0, 32, nlist)] + 0.01 * rng.standard_normal((nlist, d))
        xq = c0[rng.integers(0, 32, nq)] + 0.01 * rng.standard_normal((nq, d))
    elif kind == "offset":
        off = float(rng.choice([1.0, 5.0, 30.0])); cent = off + rng.random((nlist, d)); xq = off + rng.random((nq, d))
    elif kind == "quantised":
        cent = rng.integers(0, 4, (nlist, d)).astype(np.float64); xq = rng.integers(0, 4, (nq, d)).astype(np.float64)
    elif kind == "lowrank":
        r = max(1, d // 8); B = rng.standard_normal((r, d))
        cent = rng.standard_normal((nlist, r)) @ B + 0.05 * rng.standard_normal((nlist, d)); xq = rng.standard_normal((nq, r)) @ B
    elif kind == "tiny":
        cent = 1e-4 * rng.random((nlist, d)); xq = 1e-4 * rng.random((nq, d))
    else:
        cent = 3e3 * rng.standard_normal((nlist, d)); xq = 3e3 * rng.standard_normal((nq, d))
    cent = cent.astype(np.float32); xq = xq.astype(np.float32)
    if os.environ.get("IMI") == "1" and d >= 32 and d % 8 == 0:
        # the two halves of an inverted multi-index: sub-centroids = the halves of the first kc centroids, walk of nprobe cells
        nbits = int(rng.choice([8, 9, 10, 12])); kc = 1 << nbits; dc = d // 2
        src = np.concatenate([cent] * (kc // nlist + 1))[:kc] if nlist < kc else cent[:kc]
        imi = np.ascontiguousarray(np.stack([src[:, :dc], src[:, dc:]]))
        if nlist < kc:
            imi = imi + (1e-3 * rng.standard_normal(imi.shape)).astype(np.float32)
        imi[0, 7:20] = imi[0, 5]
        nprobe = min(nprobe, 64)
        g = vlq.GpuIVFPQ(d, kc * kc, 4, 8)
        g.set_imi_centroids(nbits, imi.astype(np.float32))
        g.set_pq_centroids(rng.random((4, 256, d // 4)).astype(np.float32))
        g.set_coarse_screen(1)
        cd1, k1 = g.coarse_search(xq, nprobe)
        st = g.coarse_screen_state()
        g.set_coarse_screen(0)
        cd0, k0 = g.coarse_search(xq, nprobe)
        ok = np.array_equal(bits(cd1), bits(cd0)) and np.array_equal(k1, k0)
        if not ok:
            bad += 1
            print("MISMATCH imi case %d: %s d=%d nbits=%d nprobe=%d nq=%d state=%s" % (case, kind, d, nbits, nprobe, nq, st), flush=True)
        elif case % 10 == 0:
            print("imi case %d ok: %s d=%d nbits=%d nprobe=%d nq=%d screen state %s (%.0f s)" % (case, kind, d, nbits, nprobe, nq, st, time.time() - t0), flush=True)
        del g
        continue
    cent[nlist // 3:nlist // 3 + 20] = cent[1]                        # duplicated centroids
    xq[:32] = cent[rng.integers(0, nlist, 32)]                       # queries on centroids
    M = 4
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(cent)
    g.set_pq_centroids(rng.random((M, 256, d // M)).astype(np.float32))
    g.set_coarse_screen(1)
    cd1, k1 = g.coarse_search(xq, nprobe)
    st = g.coarse_screen_state()
    a1 = g.coarse_search(xq, 1)
    g.set_coarse_screen(0)
    cd0, k0 = g.coarse_search(xq, nprobe)
    a0 = g.coarse_search(xq, 1)
    ok = np.array_equal(bits(cd1), bits(cd0)) and np.array_equal(k1, k0) and np.array_equal(bits(a1[0]), bits(a0[0])) and np.array_equal(a1[1], a0[1])
    if not ok:
        bad += 1
        print("MISMATCH case %d: %s d=%d nlist=%d nprobe=%d nq=%d state=%s" % (case, kind, d, nlist, nprobe, nq, st), flush=True)
    elif case % 10 == 0:
        print("case %d ok: %s d=%d nlist=%d nprobe=%d nq=%d screen state %s (%.0f s)" % (case, kind, d, nlist, nprobe, nq, st, time.time() - t0), flush=True)
    del g
print("%d cases, %d mismatches" % (ncase, bad))
sys.exit(1 if bad else 0)
