"""GPU: the list-owned scan schedule (include/vlq_ivfpq.h, vlq_ivfpq_set_scan_schedule) returns
exactly what the query-major schedule and the oracle return -- distances bit for bit, labels in the
canonical (distance, scan position) order -- over the selection classes, exact ties spread over
partitions, the max_codes cut, skipped (-1) and invalid probes, and store_pairs."""
import numpy as np
import pytest

import vector_line_quantization_amd as vlq
from oracle.pyoracle import OracleIndex
from util import bits

pytestmark = pytest.mark.gpu

import os
# 2: first build (scan16.hip OWNED), 3 / 4: second build (scan16o.hip, one / two table buffers); the whole module runs once per
# value through the parametrised fixture below
SCHEDS = [int(v) for v in os.environ.get("VLQ_TEST_OWNED_SCHEDULES", "2,3,4").split(",")]


@pytest.fixture(scope="module", params=SCHEDS)
def world(request):
    SCHED = request.param
    rng = np.random.default_rng(77)
    d, nlist, M, nb, nq = 128, 512, 16, 90000, 2500
    centres = rng.random((60, d), dtype=np.float32)
    def gen(n):
        return np.round((centres[rng.integers(0, 60, n)] + 0.06 * rng.standard_normal((n, d))) * 64).astype(np.float32)
    coarse = gen(nlist)
    pq = (rng.standard_normal((M, 256, d // M)) * 3).astype(np.float32)
    xb = gen(nb)
    xb[5000:9000] = xb[1000:5000]            # exact duplicates: equal distances in different lists' neighbours
    xq = gen(nq)
    xq[:200] = xb[1000:1200]
    ox = OracleIndex(d, nlist, M, 8, coarse, pq)
    ox.add(xb, canonical=True)
    gs = []
    for mode in (1, SCHED):
        g = vlq.GpuIVFPQ(d, nlist, M, 8)
        g.set_coarse_centroids(coarse)
        g.set_pq_centroids(pq)
        g.set_lists(ox.codes, ox.ids, ox.list_offsets)
        g.set_scan_schedule(mode)
        gs.append(g)
    return ox, gs[0], gs[1], xq


@pytest.mark.parametrize("nprobe,k", [(32, 10), (8, 1), (100, 64), (32, 100), (16, 128), (64, 300), (32, 1000)])
def test_owned_equals_query_major_and_oracle(world, nprobe, k):
    ox, g1, g2, xq = world
    g1.stats(reset=True)
    g2.stats(reset=True)
    D1, I1 = g1.search(xq, nprobe, k)
    D2, I2 = g2.search(xq, nprobe, k)
    assert np.array_equal(bits(D1), bits(D2))
    assert np.array_equal(I1, I2)
    assert g1.stats(reset=True)[1] == g2.stats(reset=True)[1]          # the same codes were visited
    pick = np.r_[0:64, 1000:1064]
    Do, Io = ox.search(xq[pick], nprobe, k, canonical=True)
    assert np.array_equal(bits(D2[pick]), bits(Do)) and np.array_equal(I2[pick], Io)


def test_owned_with_max_codes_skips_and_pairs(world):
    ox, g1, g2, xq = world
    cd, keys = g1.coarse_search(xq, 32)
    keys = keys.copy()
    keys[::3, 5] = -1                      # skipped probes (IndexIVFPQ.cpp:1004-1007)
    keys[1::7, 0] = -1
    for g in (g1, g2):
        g.set_search_options(True, 1, max_codes=3000)
    try:
        for store_pairs in (False, True):
            D1, I1 = g1.search_preassigned(xq, keys, cd, 20, store_pairs=store_pairs)
            D2, I2 = g2.search_preassigned(xq, keys, cd, 20, store_pairs=store_pairs)
            assert np.array_equal(bits(D1), bits(D2)) and np.array_equal(I1, I2)
        ox.max_codes = 3000
        Do, Io = ox.search_preassigned(xq[:100], keys[:100], cd[:100], 20, canonical=True)
        D2, I2 = g2.search_preassigned(xq, keys, cd, 20)
        assert np.array_equal(bits(D2[:100]), bits(Do)) and np.array_equal(I2[:100], Io)
    finally:
        ox.max_codes = 0
        for g in (g1, g2):
            g.set_search_options(True, 1, max_codes=0)


def test_owned_reports_invalid_keys(world):
    ox, g1, g2, xq = world
    cd, keys = g1.coarse_search(xq, 16)
    keys = keys.copy()
    keys[7, 3] = 512 + 9                   # >= nlist: the reference aborts (IndexIVFPQ.cpp:1008-1011)
    with pytest.raises(vlq.VlqError):
        g2.search_preassigned(xq, keys, cd, 5)
    g2.stats(reset=True)


def test_owned_reports_a_row_of_only_invalid_keys(world):
    """A query whose probes are ALL >= nlist has no (query, partition) item, hence no scan workgroup: the merge
    kernel raises the flag, as the query-major kernel does (the two schedules behave alike in every case)."""
    ox, g1, g2, xq = world
    cd, keys = g1.coarse_search(xq, 16)
    keys = keys.copy()
    keys[11, :] = 512 + np.arange(16)
    for g in (g1, g2):
        with pytest.raises(vlq.VlqError):
            g.search_preassigned(xq, keys, cd, 5)
        g.stats(reset=True)


def test_bad_key_flag_keeps_the_counters(world):
    """vlq_ivfpq_stats(reset = 0) consumes the flag it reports but leaves ncode alone."""
    import torch
    ox, g1, g2, xq = world
    cd, keys = g1.coarse_search(xq, 16)
    g1.stats(reset=True)
    D, I = g1.search_preassigned(xq, keys, cd, 5)
    _nq, ncode = g1.stats()
    assert ncode > 0
    bad = keys.copy()
    bad[3, 0] = 1 << 40
    xd, kd, cdd = (torch.from_numpy(a).cuda() for a in (xq, bad, cd))
    Dd = torch.empty((xq.shape[0], 5), dtype=torch.float32, device="cuda")
    Id = torch.empty((xq.shape[0], 5), dtype=torch.int64, device="cuda")
    g1.search_preassigned(xd, kd, cdd, 5, D=Dd, I=Id)      # device outputs: the error surfaces lazily
    with pytest.raises(vlq.VlqError):
        g1.stats()
    _nq, ncode2 = g1.stats()                                # flag consumed, counters kept
    assert ncode2 >= 2 * ncode - 64 * 4096
    g1.stats(reset=True)
