#!/usr/bin/env python3
"""Generate the committed golden fixtures tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference + MKL): it builds the
reference's CPU library with oracle/ref.mk, feeds deterministic numpy inputs to
oracle/_ref/ref_driver (our driver over the reference's public API) and stores
inputs + the reference's outputs as small .npz files.  The tests never call
this; they only read the .npz.  Fixtures are data (inputs and expected
outputs), no reference source text.

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py c1_small   # one case
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from tagged import read_tagged, write_tagged  # noqa: E402


def drand48_stream(seed, n):
    """libc srand48(seed); drand48() x n -- the generator the reference's own unit
    test uses (tests/test_ivfpq_indexing.cpp:44-69)."""
    a, c, mask = 0x5DEECE66D, 0xB, (1 << 48) - 1
    x = ((seed & 0xFFFFFFFF) << 16) | 0x330E
    out = np.empty(n, dtype=np.float64)
    for i in range(n):
        x = (a * x + c) & mask
        out[i] = x / float(1 << 48)
    return out


def sift_like(rng, n, d, centres, sigma):
    """Gaussian mixture, scaled to the SIFT byte range and rounded, so vectors are
    exactly representable as uint8 (keeps fixtures small)."""
    pick = rng.integers(0, centres.shape[0], size=n)
    x = centres[pick] + sigma * rng.standard_normal((n, d))
    return np.clip(np.rint(x * 255.0), 0, 255).astype(np.float32)


def gmm_case(seed, d, nc, nt, nb, nq, sigma=0.08, dup=1):
    rng = np.random.default_rng(seed)
    centres = rng.random((nc, d))
    xt = sift_like(rng, nt, d, centres, sigma)
    xb = sift_like(rng, nb // dup, d, centres, sigma)
    if dup > 1:  # exact duplicates -> exact distance ties inside lists
        xb = np.repeat(xb, dup, axis=0)
        xb = xb[rng.permutation(xb.shape[0])]
    xq = sift_like(rng, nq, d, centres, sigma)
    return xt, xb, xq


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


def cfg(d, nlist, M, nbits, nt, nb, nq, nprobe, k, max_codes=0, n_small=0,
        km_niter=0, pq_niter=0, by_residual=1, upt=-1, imi_nbits=0):
    return np.array([d, nlist, M, nbits, nt, nb, nq, nprobe, k, max_codes, n_small,
                     km_niter, pq_niter, by_residual, upt, imi_nbits], dtype=np.int64)


@case
def ref_unit_test():
    """Exactly the inputs of the reference's tests/test_ivfpq_indexing.cpp:20-100
    (d=64, 25 lists, M=16x8bit, srand48(35), nprobe=5, k=5)."""
    d, nt, nb, nq = 64, 1500, 1000, 200
    s = drand48_stream(35, (nt + nb + nq) * d).astype(np.float32)
    xt = s[:nt * d].reshape(nt, d)
    xb = s[nt * d:(nt + nb) * d].reshape(nb, d)
    xq = s[(nt + nb) * d:].reshape(nq, d)
    return cfg(d, 25, 16, 8, nt, nb, nq, 5, 5, n_small=8), xt, xb, xq, None


@case
def c1_small():
    """BASELINE config-1 shape scaled down: d=128, M=16x8bit (dsub=8), k=10."""
    xt, xb, xq = gmm_case(101, 128, 80, 6000, 5000, 64)
    return cfg(128, 64, 16, 8, 6000, 5000, 64, 8, 10, n_small=12, pq_niter=8), xt, xb, xq, None


@case
def deep_like_dsub6():
    """d=96, M=16 -> dsub=6 (SSE tail path d%4=2), 6-bit codes, all lists probed,
    k=100, custom (non-sequential) ids."""
    xt, xb, xq = gmm_case(202, 96, 50, 4000, 3000, 40)
    rng = np.random.default_rng(5)
    xids = (rng.permutation(10 ** 6)[:3000] * 7 + 3).astype(np.int64)
    return cfg(96, 37, 16, 6, 4000, 3000, 40, 37, 100, n_small=5, pq_niter=8), xt, xb, xq, xids


@case
def duplicates_ties():
    """Every database vector stored 4x -> exact ADC distance ties in the heap."""
    xt, xb, xq = gmm_case(303, 32, 20, 3000, 2000, 50, dup=4)
    return cfg(32, 16, 8, 8, 3000, 2000, 50, 6, 10, n_small=4, pq_niter=6), xt, xb, xq, None


@case
def tiny_padding():
    """Fewer stored vectors than k: results padded with -1 / FLT_MAX; some empty lists."""
    xt, xb, xq = gmm_case(404, 16, 10, 2000, 30, 25)
    return cfg(16, 20, 4, 8, 2000, 30, 25, 20, 64, n_small=3, pq_niter=4), xt, xb, xq, None


@case
def max_codes_cut():
    """max_codes early exit (IndexIVFPQ.cpp:1033)."""
    xt, xb, xq = gmm_case(505, 64, 40, 4000, 4000, 32)
    return cfg(64, 32, 8, 8, 4000, 4000, 32, 16, 20, max_codes=300, pq_niter=6), xt, xb, xq, None


@case
def table_mode0():
    """use_precomputed_table=0: per-(query,list) residual distance tables."""
    xt, xb, xq = gmm_case(606, 64, 40, 4000, 3000, 32)
    return cfg(64, 32, 16, 8, 4000, 3000, 32, 8, 10, pq_niter=6, upt=0), xt, xb, xq, None


@case
def not_by_residual():
    """by_residual=false: codes of raw vectors, one table per query."""
    xt, xb, xq = gmm_case(707, 32, 30, 3000, 2500, 30)
    return cfg(32, 24, 8, 8, 3000, 2500, 30, 6, 10, pq_niter=6, by_residual=0), xt, xb, xq, None


@case
def m32_long_codes():
    """M=32 (32-byte codes, dsub=4), 7-bit codes."""
    xt, xb, xq = gmm_case(808, 128, 60, 5000, 3000, 30)
    return cfg(128, 48, 32, 7, 5000, 3000, 30, 10, 16, pq_niter=5), xt, xb, xq, None


@case
def imi_sse_tables():
    """Inverted multi-index coarse quantizer 2 x 4 bit (256 lists), coarse sub-vectors of
    8 dims -> the reference's SSE table path (no BLAS): coarse stage bit-pinned too.
    Table mode 2 (IndexIVFPQ.cpp:430-457,645-686), M=8 -> 4 PQ sub-quantizers per half."""
    xt, xb, xq = gmm_case(909, 16, 40, 4000, 3000, 40, sigma=0.1)
    return cfg(16, 256, 8, 8, 4000, 3000, 40, 12, 10, n_small=6, km_niter=8, pq_niter=6, imi_nbits=4), xt, xb, xq, None


@case
def imi_blas_tables():
    """IMI 2 x 5 bit (1024 lists), coarse sub-vectors of 32 dims -> pairwise_L2sqr / sgemm
    tables (utils.cpp:1311-1355): coarse stage pinned to rounding only.  M=16 x 8 bit."""
    xt, xb, xq = gmm_case(1010, 64, 60, 5000, 4000, 48)
    return cfg(64, 1024, 16, 8, 5000, 4000, 48, 24, 10, km_niter=8, pq_niter=5, imi_nbits=5), xt, xb, xq, None


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def run_case(name):
    c, xt, xb, xq, xids = CASES[name]()
    subprocess.check_call(["make", "-s", "-f", "oracle/ref.mk"], cwd=ROOT)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "oracle/_ref/mkl") + ":" + env.get("LD_LIBRARY_PATH", "")
    env["OMP_NUM_THREADS"] = "4"
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        arrs = {"cfg": c, "xt": xt, "xb": xb, "xq": xq}
        if xids is not None:
            arrs["xids"] = xids
        write_tagged(fin, arrs)
        fidx = os.path.join(td, "index.faissindex")
        subprocess.check_call([os.path.join(ROOT, "oracle/_ref/ref_driver"), fin, fout, fidx], env=env)
        out = read_tagged(fout)
        index_file = np.fromfile(fidx, dtype=np.uint8)

    keep = {"cfg": c, "xq": xq}
    # inputs: store compactly when exactly byte-valued
    for nm, a in (("xb", xb),):
        if np.array_equal(a, np.rint(a)) and a.min() >= 0 and a.max() <= 255:
            keep[nm + "_u8"] = a.astype(np.uint8)
        else:
            keep[nm] = a
    if xids is not None:
        keep["xids"] = xids
    for nm in ("meta", "coarse_centroids", "imi_centroids", "pq_centroids", "list_offsets", "codes", "ids",
               "xb_assign", "keys", "coarse_dis", "D", "I", "ncode", "D_pairs", "I_pairs",
               "small_keys", "small_coarse_dis", "small_D", "small_I", "q_norms", "c_norms"):
        if nm in out:
            keep[nm] = out[nm]
    # big tables: first rows + sha256 of the whole array
    if "precomputed_table" in out:
        pt = out["precomputed_table"]
        keep["precomputed_table_head"] = pt[:3]
        keep["precomputed_table_rows"] = np.array([pt.shape[0]], np.int64)
        keep["precomputed_table_sha256"] = sha(pt)
    for nm in ("ip_table", "dis_table"):
        keep[nm + "_head"] = out[nm][:2]
        keep[nm + "_sha256"] = sha(out[nm])
    if name in ("tiny_padding", "imi_sse_tables"):   # the reference's on-disk format (write_index), as data
        keep["faissindex_file"] = index_file
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **keep)
    print("%-18s %8.1f KB  ncode=%d mode=%d" % (name, os.path.getsize(path) / 1024.0,
                                               int(out["ncode"][0]), int(out["meta"][0])))


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n)
