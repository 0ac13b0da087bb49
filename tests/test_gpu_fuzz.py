"""GPU: seeded random configurations of the IVFPQ path against the oracle, bit for bit -- shapes
the fixtures do not cover (odd d / M / nbits combinations, SSE tail lengths, empty and huge
lists, nprobe > nlist, k > ntotal, max_codes cuts, all three table modes, add in batches,
preassigned keys with -1 holes, batches on both sides of the 20-query coarse switch and of the
1024-query ordering switch)."""
import numpy as np
import pytest

import vector_line_quantization_amd as vlq
from oracle import pyoracle
from util import bits

pytestmark = pytest.mark.gpu


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    M = int(rng.choice([1, 2, 3, 4, 4, 8, 8, 12, 16, 16, 16, 20, 24, 28, 32, 32, 40, 48, 56, 64]))    # multiples of 4 (but 16) x 8 bit in table mode 1: scanm.hip
    dsub = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12]))
    if M == 16 and rng.random() < 0.5:
        dsub = int(rng.choice([6, 8]))                       # the 16-byte fast path, both table sources
    nbits = int(rng.choice([8, 8, 8, 4, 5, 6, 7])) if M != 16 or rng.random() < 0.3 else 8
    d = M * dsub
    nlist = int(rng.choice([1, 2, 7, 33, 64, 130]))
    mode = int(rng.choice([0, 1, 1, 1, 2]))                   # 2 = not by_residual
    nb = int(rng.choice([0, 5, 200, 3000, 3000]))
    nq = int(rng.choice([1, 7, 19, 20, 33, 1100, 3100]))      # 3100: the two-wave workgroups of scan16 (from 3000 queries on)
    nprobe = int(rng.choice([1, 3, 16, 64, 200]))
    k = int(rng.choice([1, 5, 64, 65, 300]))
    max_codes = int(rng.choice([0, 0, 0, 50, 400]))
    return rng, dict(M=M, dsub=dsub, nbits=nbits, d=d, nlist=nlist, mode=mode, nb=nb, nq=nq, nprobe=nprobe, k=k,
                     max_codes=max_codes)


import os


@pytest.mark.parametrize("seed", range(int(os.environ.get("VLQ_FUZZ_SEEDS", "40"))))   # VLQ_FUZZ_SEEDS=400 for a soak run
def test_random_configuration(seed):
    rng, c = draw(seed)
    d, nlist, M, nbits = c["d"], c["nlist"], c["M"], c["nbits"]
    ksub = 1 << nbits
    centres = rng.random((max(2, nlist // 3), d)).astype(np.float32)
    gen = lambda n: (centres[rng.integers(0, len(centres), n)] + 0.1 * rng.standard_normal((n, d))).astype(np.float32)
    coarse = gen(nlist)
    pq = (0.2 * rng.standard_normal((M, ksub, c["dsub"]))).astype(np.float32)
    xb, xq = gen(c["nb"]), gen(c["nq"])
    by_res, upt = c["mode"] != 2, 1 if c["mode"] == 1 else 0
    g = vlq.GpuIVFPQ(d, nlist, M, nbits)
    g.set_coarse_centroids(coarse)
    g.set_pq_centroids(pq)
    g.set_search_options(by_residual=by_res, use_precomputed_table=upt, max_codes=c["max_codes"])
    ox = pyoracle.OracleIndex(d, nlist, M, nbits, coarse, pq, by_residual=by_res, use_precomputed_table=upt,
                              max_codes=c["max_codes"])
    ids = None if seed % 3 else (rng.permutation(10 ** 6)[:c["nb"]].astype(np.int64) - 7)
    cut = c["nb"] // 3
    for a, b in ((0, cut), (cut, c["nb"])):                     # two batches (device append)
        if b > a:
            g.add(xb[a:b], None if ids is None else ids[a:b])
    if c["nb"]:
        ox.add(xb, ids, canonical=True)
    assert g.ntotal == c["nb"]
    nprobe = min(c["nprobe"], 1024)
    D, I = g.search(xq, nprobe, c["k"])
    Do, Io = ox.search(xq, nprobe, c["k"], canonical=True)
    assert np.array_equal(bits(D), bits(Do)), c
    assert np.array_equal(I, Io), c
    # the parity seam with holes in the probe lists
    cd, keys = g.coarse_search(xq, nprobe)
    keys = keys.copy()
    keys[rng.random(keys.shape) < 0.2] = -1
    D2, I2 = g.search_preassigned(xq, keys, cd, c["k"], store_pairs=bool(seed & 1))
    Do2, Io2 = ox.search_preassigned(xq, keys, cd, c["k"], store_pairs=bool(seed & 1), canonical=True)
    assert np.array_equal(bits(D2), bits(Do2)), c
    assert np.array_equal(I2, Io2), c


@pytest.mark.parametrize("k,dup", [(10, 1), (64, 1), (100, 1), (100, 4), (128, 1), (200, 1), (256, 4), (400, 1), (1000, 1)])
def test_long_lists_every_k_class(k, dup):
    """Mean list length >= 1024: the pipelined two-chunk loop of scan16 for k <= 64 and, with the
    `long_lists` switch, for the 128- and 256-key selections; larger k keeps the plain loop.  `dup`
    stores every vector several times (exact distance ties across chunks and waves)."""
    rng = np.random.default_rng(77 + k + dup)
    d, nlist, M, nb, nq, nprobe = 32, 6, 16, 12000, 40, 4
    centres = rng.random((3, d)).astype(np.float32)
    gen = lambda n: (centres[rng.integers(0, 3, n)] + 0.1 * rng.standard_normal((n, d))).astype(np.float32)
    coarse = gen(nlist)
    pq = (0.2 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    xb = np.repeat(gen(nb // dup), dup, axis=0)[rng.permutation(nb // dup * dup)]
    xq = gen(nq)
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(coarse)
    g.set_pq_centroids(pq)
    ox = pyoracle.OracleIndex(d, nlist, M, 8, coarse, pq)
    g.add(xb)
    ox.add(xb, None, canonical=True)
    assert g.ntotal >= 1024 * nlist and max(g.list_length(i) for i in range(nlist)) > 2048
    D, I = g.search(xq, nprobe, k)
    Do, Io = ox.search(xq, nprobe, k, canonical=True)
    assert np.array_equal(bits(D), bits(Do))
    assert np.array_equal(I, Io)


@pytest.mark.parametrize("k,dup", [(129, 1), (200, 8), (256, 1), (257, 16), (512, 40), (700, 1), (1024, 8)])
def test_medium_lists_shared_queue_selection(k, dup):
    """Lists of 24 .. 1023 codes on average and k > 128: scan16_bigk_kernel (capacities 256 / 512 / 1024) -- one
    selection per workgroup, the k-th key of a flush found by band-relative bucket counts.  `dup` stores every
    vector several times: runs of equal distances longer than a 64-key band (the gathered last step), flushes
    with fewer than k real keys, k > the codes a query sees (padding rows)."""
    rng = np.random.default_rng(500 + k + dup)
    d, nlist, M, nb, nq, nprobe = 32, 24, 16, 9600, 48, 9
    centres = rng.random((5, d)).astype(np.float32)
    gen = lambda n: (centres[rng.integers(0, 5, n)] + 0.1 * rng.standard_normal((n, d))).astype(np.float32)
    coarse = gen(nlist)
    pq = (0.2 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    xb = np.repeat(gen(nb // dup), dup, axis=0)[rng.permutation(nb // dup * dup)]
    xq = gen(nq)
    xq[:8] = xb[:8]
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(coarse)
    g.set_pq_centroids(pq)
    ox = pyoracle.OracleIndex(d, nlist, M, 8, coarse, pq)
    g.add(xb)
    ox.add(xb, None, canonical=True)
    assert 24 * nlist <= g.ntotal < 1024 * nlist
    for np_ in (nprobe, 1):                                   # nprobe 1: fewer codes than k for the small lists
        D, I = g.search(xq, np_, k)
        Do, Io = ox.search(xq, np_, k, canonical=True)
        assert np.array_equal(bits(D), bits(Do))
        assert np.array_equal(I, Io)
