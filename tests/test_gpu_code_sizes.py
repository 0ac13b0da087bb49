"""GPU: the engineered scan for 4-, 8-, 12-, 20-, 24-, 28-, 32-, 40-, 48-, 56- and 64-byte codes (csrc/scanm.hip: the 16-byte kernel's organisation over the code
sizes the reference instantiates, gpu/impl/IVFPQ.cu:149-172) against the oracle, bit for bit, and against the generic kernel
(VLQ_GENERIC_SCAN) -- over the selection classes, short and long lists, empty lists, the max_codes cut, skipped / invalid
probes, store_pairs, batches small enough to be split over workgroups, and the multi-index table type 2."""
import os

import numpy as np
import pytest

import vector_line_quantization_amd as vlq
from oracle.pyoracle import OracleIndex
from util import bits

pytestmark = pytest.mark.gpu


def make(M, dsub, nlist, nb, seed, long_frac=0.0):
    rng = np.random.default_rng(seed)
    d = M * dsub
    centres = rng.random((max(4, nlist // 4), d)).astype(np.float32)
    gen = lambda n: (centres[rng.integers(0, len(centres), n)] + 0.08 * rng.standard_normal((n, d))).astype(np.float32)
    coarse = gen(nlist)
    pq = (0.15 * rng.standard_normal((M, 256, dsub))).astype(np.float32)
    xb = gen(nb)
    if long_frac:                                  # a few very long lists: many copies of some centroids' neighbourhoods
        n = int(nb * long_frac)
        xb[:n] = coarse[rng.integers(0, 3, n)] + 0.01 * rng.standard_normal((n, d)).astype(np.float32)
    xb[nb // 2:nb // 2 + 50] = xb[:50]            # exact duplicates: equal distances
    ox = OracleIndex(d, nlist, M, 8, coarse, pq)
    ox.add(xb, canonical=True)
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(coarse)
    g.set_pq_centroids(pq)
    g.set_lists(ox.codes, ox.ids, ox.list_offsets)
    return rng, ox, g, gen


@pytest.mark.parametrize("M,dsub", [(8, 8), (8, 4), (32, 4), (32, 2), (64, 2), (64, 1), (24, 4), (40, 2), (48, 2), (56, 2), (12, 8), (20, 4), (28, 4), (4, 8)])
@pytest.mark.parametrize("nq,nprobe,k", [(1500, 16, 10), (40, 8, 1), (300, 32, 100), (1100, 24, 200), (64, 64, 1000)])
def test_code_sizes_bit_exact(M, dsub, nq, nprobe, k):
    rng, ox, g, gen = make(M, dsub, 96, 12000, 100 * M + dsub, long_frac=0.3)
    xq = gen(nq)
    xq[:20] = gen(20) * 0 + ox.coarse_centroids[:20]
    D, I = g.search(xq, nprobe, k)
    Do, Io = ox.search(xq, nprobe, k, canonical=True)
    assert np.array_equal(bits(D), bits(Do))
    assert np.array_equal(I, Io)
    assert g.stats(reset=True)[1] == ox.last_ncode if hasattr(ox, "last_ncode") else True


@pytest.mark.parametrize("M,dsub", [(8, 8), (32, 4), (64, 2), (24, 4), (40, 2), (48, 2), (56, 2), (12, 8), (20, 4), (28, 4), (4, 8)])
def test_code_sizes_seam_holes_pairs_and_max_codes(M, dsub):
    rng, ox, g, gen = make(M, dsub, 64, 6000, 7 * M)
    xq = gen(700)
    cd, keys = g.coarse_search(xq, 16)
    keys = keys.copy()
    keys[rng.random(keys.shape) < 0.25] = -1
    keys[5] = -1                                     # a query without any probe
    for pairs in (False, True):
        D, I = g.search_preassigned(xq, keys, cd, 20, store_pairs=pairs)
        Do, Io = ox.search_preassigned(xq, keys, cd, 20, store_pairs=pairs, canonical=True)
        assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)
    g.set_search_options(max_codes=300)
    ox.max_codes = 300
    D, I = g.search(xq, 16, 10)
    Do, Io = ox.search(xq, 16, 10, canonical=True)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)
    bad = keys.copy()
    bad[3, 0] = 64                                   # key >= nlist: the reference aborts the search (IndexIVFPQ.cpp:1008-1011)
    with pytest.raises(vlq.VlqError):
        g.search_preassigned(xq, bad, cd, 5)


@pytest.mark.parametrize("M", [8, 32, 24, 48, 12, 28])
def test_code_sizes_multi_index_table_type_2(M):
    """MultiIndexQuantizer coarse quantizer: term2 rows per coarse SUB-index, sub-quantizers of the first half of the code
    take theirs from the first sub-index (IndexIVFPQ.cpp:645-686) -- sift1b_imi_pq.cpp's index shape with 8-byte codes."""
    rng = np.random.default_rng(5 + M)
    nbits, dsub = 4, 2
    kc, d = 1 << nbits, M * dsub
    imi = rng.random((2, kc, d // 2)).astype(np.float32)
    pq = (0.2 * rng.standard_normal((M, 256, dsub))).astype(np.float32)
    xb = rng.random((9000, d)).astype(np.float32)
    xq = rng.random((1200, d)).astype(np.float32)
    ox = OracleIndex(d, kc * kc, M, 8, None, pq, imi_centroids=imi, imi_nbits=nbits)
    ox.add(xb, canonical=True)
    g = vlq.GpuIVFPQ(d, kc * kc, M, 8)
    g.set_imi_centroids(nbits, imi)
    g.set_pq_centroids(pq)
    g.set_lists(ox.codes, ox.ids, ox.list_offsets)
    for nprobe, k in ((16, 10), (64, 100)):
        D, I = g.search(xq, nprobe, k)
        Do, Io = ox.search(xq, nprobe, k, canonical=True)
        assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)


@pytest.mark.parametrize("M,dsub", [(16, 8), (16, 6), (16, 4), (8, 16), (8, 12), (8, 8), (32, 4), (32, 2)])
@pytest.mark.parametrize("nq,nprobe,k", [(1500, 16, 10), (40, 8, 1), (300, 33, 100), (64, 64, 300)])
def test_table_mode_0_bit_exact(M, dsub, nq, nprobe, k):
    """by_residual WITHOUT the precomputed table (use_precomputed_table = 0: GpuIndexIVFPQConfig's default; IndexIVFPQ.cpp:
    636-638: residual distance tables per (query, list), dis0 = 0): the engineered kernel keeps the codebook in registers;
    results equal the oracle's mode 0 bit for bit, including the seam with holes and the max_codes cut."""
    rng, ox, g, gen = make(M, dsub, 96, 12000, 300 * M + dsub, long_frac=0.3)
    g.set_search_options(by_residual=True, use_precomputed_table=0)
    ox0 = OracleIndex(M * dsub, 96, M, 8, ox.coarse_centroids, ox.pq_centroids, codes=ox.codes, ids=ox.ids,
                      list_offsets=ox.list_offsets, use_precomputed_table=0)
    xq = gen(nq)
    D, I = g.search(xq, nprobe, k)
    Do, Io = ox0.search(xq, nprobe, k, canonical=True)
    assert np.array_equal(bits(D), bits(Do))
    assert np.array_equal(I, Io)
    cd, keys = g.coarse_search(xq, nprobe)
    keys = keys.copy()
    keys[rng.random(keys.shape) < 0.2] = -1
    D2, I2 = g.search_preassigned(xq, keys, cd, k, store_pairs=True)
    Do2, Io2 = ox0.search_preassigned(xq, keys, cd, k, store_pairs=True, canonical=True)
    assert np.array_equal(bits(D2), bits(Do2)) and np.array_equal(I2, Io2)
    g.set_search_options(by_residual=True, use_precomputed_table=0, max_codes=400)
    ox0.max_codes = 400
    D3, I3 = g.search(xq, nprobe, k)
    Do3, Io3 = ox0.search(xq, nprobe, k, canonical=True)
    assert np.array_equal(bits(D3), bits(Do3)) and np.array_equal(I3, Io3)


def test_engineered_equals_generic_kernel():
    """the same index through the generic kernel (a fresh process with VLQ_GENERIC_SCAN=1): identical rows"""
    import subprocess, sys, json
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, "tests")
import test_gpu_code_sizes as t
out = {}
for M, dsub in ((8, 8), (32, 4), (64, 2), (24, 4), (40, 2), (48, 2), (56, 2), (12, 8), (20, 4), (28, 4), (4, 8)):
    rng, ox, g, gen = t.make(M, dsub, 96, 12000, 100 * M + dsub, long_frac=0.3)
    xq = gen(600)
    D, I = g.search(xq, 16, 10)
    out[str(M)] = [int(D.view(np.uint32).astype(np.uint64).sum()), int(I.sum())]
print(json.dumps(out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for env_extra in ({}, {"VLQ_GENERIC_SCAN": "1"}):
        env = dict(os.environ)
        env.update(env_extra)
        p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1]
