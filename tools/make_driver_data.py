#!/usr/bin/env python3
"""Synthetic input files for the REFERENCE's own drivers (tests/demo_sift1M.cpp reads
/home/data/sift1m/{learn,base,query}.fvecs + groundtruth.ivecs): generator G1 of SURVEY.md section 8(d) with
real-valued noise (no exact distance ties, so CPU and device runs pick the same neighbours), exact ground
truth by brute force.  The drivers' hard-coded /home/data prefix is redirected to <root> by the interposer
(VLQ_DATA_ROOT).      python tools/make_driver_data.py <root> [nt nb nq]
                      python tools/make_driver_data.py <root> sift1b <cwd> [nb nq]
                      python tools/make_driver_data.py <root> deep1b <cwd> [nb nq]   (tests/deep1b_imi_pq.cpp, deep1b16_imi_pq.cpp)
                      ... [nb nq populated]: also the drivers' POPULATED cache (sift1b: 1; deep1b: 8 or 16 = the code size), so that
                      a test needs no populating run; deep1b -8 / -16: only that driver's trained cache (it populates itself)
The second form prepares tests/sift1b_imi_pq.cpp: byte-valued base.umem / query.umem / learn.umem ("num dim" text
header, data from byte 20, :100-150), gnd/idx_1000M.ivecs, and -- in <cwd>, where the driver looks for its cache
(:225-243) -- sift1b_14_8_trained_index.faissindex in the reference's file format (index_io.cpp:226-317): the
driver's own training step is 2 M vectors of k-means into 2 x 16 384 centroids on the host, hours of CPU; the
cached-index branch of the driver skips it.  The trained state written here (inverted multi-index 2 x 14 bits with
sampled sub-centroids, 8 x 8-bit PQ sampled from residuals) is as good as any for what the test checks: the
device run and the CPU-only run of the same binary must answer alike."""
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


def write_imi_ivfpq_index(path, d, imi_nbits, imi_centroids, M, pq_centroids, lists=None, nprobe=1):
    """IndexIVFPQ over a MultiIndexQuantizer 2 x imi_nbits in the reference's file format (index_io.cpp:226-317:
    "IvPQ" header, nlist, nprobe, nested "Imiq" quantizer, one id vector per list, direct map, by_residual,
    code_size, PQ, one code vector per list).  lists: None = all lists empty, else (codes[n][M] u8, ids[n] i64,
    offsets[nlist+1])."""
    import struct
    nlist = 1 << (2 * imi_nbits)
    hdr = lambda f, dd, ntotal: f.write(struct.pack("<iqqq?i", dd, ntotal, 1 << 20, 1 << 20, True, 1))

    def vec(f, a):
        a = np.ascontiguousarray(a)
        f.write(struct.pack("<Q", a.size))
        f.write(a.tobytes())

    def pqrec(f, dd, MM, cent):
        f.write(struct.pack("<QQQ", dd, MM, 8 if MM != 2 else imi_nbits))
        vec(f, np.asarray(cent, np.float32))

    def list_vectors(f, width, payload, offsets):
        if offsets is None:        # every list: an 8-byte zero length
            z = np.zeros(1 << 20, np.uint64)
            for _ in range(nlist >> 20):
                z.tofile(f)
            np.zeros(nlist & ((1 << 20) - 1), np.uint64).tofile(f)
            return
        if isinstance(offsets, tuple):      # sparse: (sorted list ids of the non-empty lists, their lengths, starts in payload)
            lid, ln, st = offsets
            z = np.zeros(1 << 20, np.uint64)
            prev = 0

            def zeros(cnt):
                while cnt > 0:
                    c = min(cnt, z.size)
                    z[:c].tofile(f)
                    cnt -= c
            for i in range(lid.size):
                zeros(int(lid[i]) - prev)
                vec(f, payload[int(st[i]) * width:(int(st[i]) + int(ln[i])) * width])
                prev = int(lid[i]) + 1
            zeros(nlist - prev)
            return
        for i in range(nlist):
            vec(f, payload[offsets[i] * width:offsets[i + 1] * width])

    ntotal = 0 if lists is None else (int(lists[2][1].sum()) if isinstance(lists[2], tuple) else int(lists[2][-1]))
    with open(path, "wb") as f:
        f.write(b"IvPQ")
        hdr(f, d, ntotal)
        f.write(struct.pack("<QQ", nlist, nprobe))
        f.write(b"Imiq")
        hdr(f, d, nlist)
        pqrec(f, d, 2, imi_centroids)
        list_vectors(f, 1, None if lists is None else np.asarray(lists[1], np.int64), None if lists is None else lists[2])
        f.write(struct.pack("<?", False))
        vec(f, np.zeros(0, np.int64))
        f.write(struct.pack("<?Q", True, M))
        pqrec(f, d, M, pq_centroids)
        list_vectors(f, M, None if lists is None else np.asarray(lists[0], np.uint8).reshape(-1), None if lists is None else lists[2])


def populated_lists(xf, imi, pq, nbits, M):
    """what the driver's add loop leaves in the lists (IndexIVFPQ::add_core_o, IndexIVFPQ.cpp:192-272): list = the nearest
    sub-centroid of each half (i0 | i1 << nbits), code = the nearest PQ centroid of the residual per sub-quantizer, ids 0 .. n-1
    in list order -- computed here so that the test starts from the driver's CACHED populated index (both runs equally cold)
    instead of spending a CPU-only run on building it.  Returns (codes, ids, (list ids, lengths, starts))."""
    n, d = xf.shape
    hd, ds = d // 2, d // M
    key = np.zeros(n, np.int64)
    res = np.empty_like(xf)
    for h in range(2):
        sl = slice(h * hd, (h + 1) * hd)
        a = np.empty(n, np.int64)
        cn = (imi[h] ** 2).sum(1)
        for i in range(0, n, 8192):
            a[i:i + 8192] = (cn[None, :] - 2.0 * xf[i:i + 8192, sl] @ imi[h].T).argmin(1)
        key |= a << (nbits * h)
        res[:, sl] = xf[:, sl] - imi[h][a]
    codes = np.empty((n, M), np.uint8)
    for m in range(M):
        sl = slice(m * ds, (m + 1) * ds)
        codes[:, m] = (((pq[m] ** 2).sum(1))[None, :] - 2.0 * res[:, sl] @ pq[m].T).argmin(1)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    lid, st, ln = np.unique(ks, return_index=True, return_counts=True)
    return codes[order], order.astype(np.int64), (lid, ln, st)


def umem_write(path, x_u8):
    n, d = x_u8.shape
    with open(path, "wb") as f:
        f.write(("%d %d\n" % (n, d)).encode().ljust(20, b" "))
        np.ascontiguousarray(x_u8, np.uint8).tofile(f)


def sift1b(root, cwd, nb=500000, nq=1000, populated=0):
    d, nc, sigma, rank, spread, kgt = 128, 2000, 0.005, 12, 0.4, 100
    centres = np.random.default_rng(1).random((nc, d)).astype(np.float32)
    sub = (np.random.default_rng(2).standard_normal((rank, d)) / np.sqrt(rank)).astype(np.float32)

    def gen(seed, n):
        r = np.random.default_rng(seed)
        x = centres[r.integers(0, nc, n)] + sigma * r.standard_normal((n, d)) + spread * r.standard_normal((n, rank)) @ sub
        return np.clip(np.rint(128.0 + 60.0 * (x - 0.5)), 0, 255).astype(np.uint8)

    os.makedirs(os.path.join(root, "sift1b", "gnd"), exist_ok=True)
    os.makedirs(cwd, exist_ok=True)
    xb, xq = gen(22, nb), gen(33, nq)
    umem_write(os.path.join(root, "sift1b", "base.umem"), xb)
    umem_write(os.path.join(root, "sift1b", "query.umem"), xq)
    umem_write(os.path.join(root, "sift1b", "learn.umem"), gen(11, 1000))     # read, never used: the cached index is
    xbf, xqf = xb.astype(np.float64), xq.astype(np.float64)
    bn = (xbf ** 2).sum(1)
    gt = np.empty((nq, kgt), np.int32)
    for i in range(0, nq, 256):
        dist = bn[None, :] - 2.0 * xqf[i:i + 256] @ xbf.T
        idx = np.argpartition(dist, kgt, axis=1)[:, :kgt]
        o = np.argsort(np.take_along_axis(dist, idx, 1), axis=1, kind="stable")
        gt[i:i + 256] = np.take_along_axis(idx, o, 1)
    ivecs_write(os.path.join(root, "sift1b", "gnd", "idx_1000M.ivecs"), gt)
    # the trained state the driver's cache branch loads: 2 x 16 384 sub-centroids sampled from the data's halves
    # (de-duplicated by a tiny jitter), 8 x 256 PQ centroids sampled from residuals to them
    r = np.random.default_rng(5)
    nbits, kc, M = 14, 1 << 14, 8
    tr = gen(44, 60000).astype(np.float32)
    imi = np.stack([tr[r.permutation(tr.shape[0])[:kc], h * 64:(h + 1) * 64] for h in range(2)])
    imi = (imi + 0.25 * r.standard_normal(imi.shape)).astype(np.float32)
    res = np.empty_like(tr[:20000])
    for h in range(2):
        sl = slice(h * 64, (h + 1) * 64)
        a = (((imi[h] ** 2).sum(1))[None, :] - 2.0 * tr[:20000, sl] @ imi[h].T).argmin(1)
        res[:, sl] = tr[:20000, sl] - imi[h][a]
    pq = np.stack([res[r.permutation(20000)[:256], m * 16:(m + 1) * 16] for m in range(M)]).astype(np.float32)
    if populated:
        # the driver loads the trained cache only to replace it by the populated one (:238-262): a 2 x 1-bit stand-in
        write_imi_ivfpq_index(os.path.join(cwd, "sift1b_14_8_trained_index.faissindex"), d, 1, imi[:, :2], M, pq)
        write_imi_ivfpq_index(os.path.join(cwd, "sift1b_14_8_populated_index.faissindex"), d, nbits, imi, M, pq,
                              lists=populated_lists(xb.astype(np.float32), imi, pq, nbits, M))
    else:
        write_imi_ivfpq_index(os.path.join(cwd, "sift1b_14_8_trained_index.faissindex"), d, nbits, imi, M, pq)
    print("wrote %s/sift1b (base %d, query %d, ground truth %d) and %s/sift1b_14_8_{trained,populated}_index.faissindex" % (root, nb, nq, kgt, cwd))


def mem_write(path, x, dtype):
    """the fork's .umem / .imem container (filehelper.cpp:253-342): the text "<num> <dim>" in the first 20 bytes, rows behind"""
    n, d = x.shape
    with open(path, "wb") as f:
        f.write(("%d %d\n" % (n, d)).encode().ljust(20, b" "))
        np.ascontiguousarray(x, dtype).tofile(f)


def deep1b(root, cwd, nb=500000, nq=1000, populated=0):
    """tests/deep1b_imi_pq.cpp and tests/deep1b16_imi_pq.cpp (BASELINE configs[3] / [4] name them): float rows of 96
    dimensions in deep1B/{learn,base,query}.umem, ground truth in deep1B/truth.imem, and in <cwd> the trained-index caches
    both drivers look for (:246-256): deep1b_14_8_ / deep1b_14_16_trained_index.faissindex (multi-index 2 x 14 bits over
    48-dimensional halves, 8 / 16 PQ bytes)."""
    d, nc, sigma, rank, spread, kgt = 96, 2000, 0.005, 12, 0.4, 100
    centres = np.random.default_rng(1).random((nc, d)).astype(np.float32)
    sub = (np.random.default_rng(2).standard_normal((rank, d)) / np.sqrt(rank)).astype(np.float32)

    def gen(seed, n):
        r = np.random.default_rng(seed)
        x = centres[r.integers(0, nc, n)] + sigma * r.standard_normal((n, d)) + spread * r.standard_normal((n, rank)) @ sub
        return (x - 0.5).astype(np.float32)

    os.makedirs(os.path.join(root, "deep1B"), exist_ok=True)
    os.makedirs(cwd, exist_ok=True)
    xb, xq = gen(22, nb), gen(33, nq)
    mem_write(os.path.join(root, "deep1B", "base.umem"), xb, np.float32)
    mem_write(os.path.join(root, "deep1B", "query.umem"), xq, np.float32)
    mem_write(os.path.join(root, "deep1B", "learn.umem"), gen(11, 1000), np.float32)   # read, never used: the index is cached
    xbf, xqf = xb.astype(np.float64), xq.astype(np.float64)
    bn = (xbf ** 2).sum(1)
    gt = np.empty((nq, kgt), np.int32)
    for i in range(0, nq, 256):
        dist = bn[None, :] - 2.0 * xqf[i:i + 256] @ xbf.T
        idx = np.argpartition(dist, kgt, axis=1)[:, :kgt]
        o = np.argsort(np.take_along_axis(dist, idx, 1), axis=1, kind="stable")
        gt[i:i + 256] = np.take_along_axis(idx, o, 1)
    mem_write(os.path.join(root, "deep1B", "truth.imem"), gt, np.int32)
    r = np.random.default_rng(5)
    nbits, kc, hd = 14, 1 << 14, d // 2
    tr = gen(44, 60000)
    imi = np.stack([tr[r.permutation(tr.shape[0])[:kc], h * hd:(h + 1) * hd] for h in range(2)])
    imi = (imi + 0.002 * r.standard_normal(imi.shape)).astype(np.float32)
    res = np.empty_like(tr[:20000])
    for h in range(2):
        sl = slice(h * hd, (h + 1) * hd)
        a = (((imi[h] ** 2).sum(1))[None, :] - 2.0 * tr[:20000, sl] @ imi[h].T).argmin(1)
        res[:, sl] = tr[:20000, sl] - imi[h][a]
    for M in (8, 16):
        ds = d // M
        pq = np.stack([res[r.permutation(20000)[:256], m * ds:(m + 1) * ds] for m in range(M)]).astype(np.float32)
        if populated and abs(populated) != M:
            continue               # (populated = 8 / 16: only that driver's files; -8 / -16: only that driver's TRAINED cache)
        if populated > 0:
            write_imi_ivfpq_index(os.path.join(cwd, "deep1b_14_%d_trained_index.faissindex" % M), d, 1, imi[:, :2], M, pq)
            write_imi_ivfpq_index(os.path.join(cwd, "deep1b_14_%d_populated_index.faissindex" % M), d, nbits, imi, M, pq,
                                  lists=populated_lists(xb, imi, pq, nbits, M))
            continue
        write_imi_ivfpq_index(os.path.join(cwd, "deep1b_14_%d_trained_index.faissindex" % M), d, nbits, imi, M, pq)
    print("wrote %s/deep1B (base %d, query %d, ground truth %d) and %s/deep1b_14_{8,16}_trained_index.faissindex" % (root, nb, nq, kgt, cwd))


def main():
    root = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "sift1b":
        return sift1b(root, sys.argv[3], *(int(v) for v in sys.argv[4:7]))
    if len(sys.argv) > 2 and sys.argv[2] == "deep1b":
        return deep1b(root, sys.argv[3], *(int(v) for v in sys.argv[4:7]))
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
