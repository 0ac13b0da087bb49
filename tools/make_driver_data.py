#!/usr/bin/env python3
"""Synthetic input files for the REFERENCE's own drivers (tests/demo_sift1M.cpp reads
/home/data/sift1m/{learn,base,query}.fvecs + groundtruth.ivecs): generator G1 of SURVEY.md section 8(d) with
real-valued noise (no exact distance ties, so CPU and device runs pick the same neighbours), exact ground
truth by brute force.  The drivers' hard-coded /home/data prefix is redirected to <root> by the interposer
(VLQ_DATA_ROOT).      python tools/make_driver_data.py <root> [nt nb nq]"""
import os
import sys

import numpy as np


def fvecs_write(path, x):
    x = np.ascontiguousarray(x, np.float32)
    n, d = x.shape
    out = np.empty((n, d + 1), np.float32)
    out[:, 0] = np.int32(d).view(np.float32)
    out[:, 1:] = x
    out.tofile(path)


def ivecs_write(path, x):
    x = np.ascontiguousarray(x, np.int32)
    n, d = x.shape
    out = np.empty((n, d + 1), np.int32)
    out[:, 0] = d
    out[:, 1:] = x
    out.tofile(path)


def main():
    root = sys.argv[1]
    nt, nb, nq = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (40000, 200000, 1000)
    # low intrinsic dimension (what makes real descriptors rankable by short codes, bench.py's second data set):
    # most of a point's offset from its centre lies in one fixed 12-dimensional subspace
    d, nc, sigma, rank, spread, kgt = 128, 2000, 0.005, 12, 0.4, 100
    centres = np.random.default_rng(1).random((nc, d)).astype(np.float32)
    sub = (np.random.default_rng(2).standard_normal((rank, d)) / np.sqrt(rank)).astype(np.float32)

    def gen(seed, n):
        r = np.random.default_rng(seed)
        x = centres[r.integers(0, nc, n)] + sigma * r.standard_normal((n, d)) + spread * r.standard_normal((n, rank)) @ sub
        return (255.0 * x).astype(np.float32)

    os.makedirs(os.path.join(root, "sift1m"), exist_ok=True)
    xt, xb, xq = gen(11, nt), gen(22, nb), gen(33, nq)
    bn = (xb.astype(np.float64) ** 2).sum(1)
    gt = np.empty((nq, kgt), np.int32)
    for i in range(0, nq, 256):
        q = xq[i:i + 256].astype(np.float64)
        dist = bn[None, :] - 2.0 * q @ xb.astype(np.float64).T
        idx = np.argpartition(dist, kgt, axis=1)[:, :kgt]
        o = np.argsort(np.take_along_axis(dist, idx, 1), axis=1, kind="stable")
        gt[i:i + 256] = np.take_along_axis(idx, o, 1)
    fvecs_write(os.path.join(root, "sift1m", "learn.fvecs"), xt)
    fvecs_write(os.path.join(root, "sift1m", "base.fvecs"), xb)
    fvecs_write(os.path.join(root, "sift1m", "query.fvecs"), xq)
    ivecs_write(os.path.join(root, "sift1m", "groundtruth.ivecs"), gt)
    print("wrote %s/sift1m: learn %d, base %d, query %d x %d, ground truth %d" % (root, nt, nb, nq, d, kgt))


if __name__ == "__main__":
    main()
