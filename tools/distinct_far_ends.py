#!/usr/bin/env python3
"""How many DISTINCT far-end centroids do the w1 kept lines of a VLQ query hold?  (numpy restatement of the line
select, sumAlongRowsWithOrder2, on a synthetic centroid set; DESIGN.md section 3, VLQ, round 3.)
   python tools/distinct_far_ends.py <nlist> <nedge> uniform|<rank>      e.g. 65536 64 uniform ; 65536 64 16"""
import numpy as np, sys, time
rng = np.random.default_rng(0)
nlist, d, nedge, nprobe, w1 = int(sys.argv[1]), 96, int(sys.argv[2]), 64, 1024
mode = sys.argv[3]
if mode == "uniform":
    cent = rng.random((nlist, d), dtype=np.float32)
else:  # low intrinsic dimension: rank-r latent + noise
    r = int(mode)
    B = rng.standard_normal((r, d)).astype(np.float32)
    cent = (rng.standard_normal((nlist, r)).astype(np.float32) @ B) / np.sqrt(r) + 0.05 * rng.standard_normal((nlist, d)).astype(np.float32)
cn = (cent**2).sum(1)
t0 = time.time()
ei = np.empty((nlist, nedge), np.int32); ed = np.empty((nlist, nedge), np.float32)
for i in range(0, nlist, 4096):
    D = cn[i:i+4096, None] + cn[None, :] - 2 * cent[i:i+4096] @ cent.T
    D[np.arange(D.shape[0]), np.arange(i, i + D.shape[0])] = np.inf
    idx = np.argpartition(D, nedge, axis=1)[:, :nedge]
    dd = np.take_along_axis(D, idx, 1); o = np.argsort(dd, 1)
    ei[i:i+4096] = np.take_along_axis(idx, o, 1); ed[i:i+4096] = np.take_along_axis(dd, o, 1)
print("graph", time.time() - t0)
nq = 64
pick = rng.integers(0, nlist, nq)
if mode == "uniform":
    xq = cent[pick] + 0.08 * rng.standard_normal((nq, d)).astype(np.float32)
else:
    xq = cent[pick] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)
Dq = (xq**2).sum(1)[:, None] + cn[None, :] - 2 * xq @ cent.T
res = []
for q in range(nq):
    row = Dq[q]
    anchors = np.argsort(row)[:nprobe]
    c = np.repeat(anchors, nedge); s = ei[anchors].ravel(); c2 = ed[anchors].ravel()
    a2 = row[s]; b2 = row[c]; t = a2 - b2 - c2
    key = np.where(t > 0, b2, b2 - 0.25 * t * t / c2)
    keep = np.argsort(key, kind="stable")[:w1]
    ks, kc = s[keep], c[keep]
    ds = np.unique(ks); dc = np.unique(kc)
    s_in_anchor = np.isin(ds, anchors).sum()
    nodes = np.unique(np.r_[ks, kc]).size
    # undirected duplicate lines (c,s) and (s,c)
    und = np.unique(np.sort(np.c_[kc, ks], 1), axis=0).shape[0]
    res.append((ds.size, dc.size, s_in_anchor, nodes, und))
res = np.array(res)
print("distinct s %.0f  distinct c %.0f  s also a probed anchor %.0f  distinct nodes %.0f  undirected lines %.0f" % tuple(res.mean(0)))
