"""GPU: the multi-index coarse stage beyond 64 cells per query (csrc/imi_wide.hip) against the oracle, bit for bit.  The
reference's MultiIndexQuantizer::search has no limit on k (IndexPQ.cpp:804-857) and its drivers ask for 2048 cells
(tests/sift1b_imi_pq.cpp:363): the sorted T = min(k, 2^nbits) smallest entries of each half table by the radix select
(T > 1024; WaveSelect below), the MinSumK replay (IndexPQ.cpp:690-778) by one wave per query with its heap in LDS (k > 128),
and a whole search whose 2048 probes are scanned in two runs and joined.  Data on an integer grid tie in long runs -- in the
tables (the T-th rank falls inside a run of equal values: the select's column-order tail) and in the sums (the heap's
positions decide, Heap.h:89-127)."""
import numpy as np
import pytest

import vector_line_quantization_amd as vlq
from oracle.pyoracle import OracleIndex
from util import bits, assert_same_topk

pytestmark = pytest.mark.gpu


def _draw(rng, grid):
    if grid >= 1000:
        return lambda shape: rng.random(shape, dtype=np.float32)
    return lambda shape: rng.integers(0, grid, shape).astype(np.float32)


def _pair(nbits, dc, grid, nq, seed):
    rng = np.random.default_rng(seed)
    kc, d, M = 1 << nbits, 2 * dc, 2
    draw = _draw(rng, grid)
    imi = draw((2, kc, dc))
    pq = rng.random((M, 256, d // M), dtype=np.float32)
    xq = draw((nq, d))
    m = min(8, kc)
    xq[:m] = np.concatenate([imi[0, :m], imi[1, :m]], axis=1)          # queries ON cells: zeros and ties at the front
    g = vlq.GpuIVFPQ(d, kc * kc, M, 8)
    g.set_imi_centroids(nbits, imi)
    g.set_pq_centroids(pq)
    ox = OracleIndex(d, kc * kc, M, 8, None, pq, imi_centroids=imi, imi_nbits=nbits)
    return g, ox, xq


@pytest.mark.parametrize("nbits,dc,grid,nq,probes", [
    (11, 4, 1000, 200, (65, 128, 129, 200, 1000, 1024, 1025, 1500, 2048, 3000, 4096)),     # fvec_L2sqr tables (dsub < 16)
    (11, 4, 6, 200, (129, 1024, 1025, 1500, 2047, 2048, 4096)),                            # integer tables: ties everywhere
    (12, 16, 1000, 120, (2048, 2049, 4096)),                                               # pairwise_L2sqr form (dsub >= 16)
    (12, 16, 3, 120, (1500, 2048, 4096)),
    (6, 8, 3, 200, (100, 129, 1000, 2048, 4096)),                                          # 64-entry tables: every cell at 4096
    (6, 8, 1000, 200, (130, 4095, 4096)),
    (14, 8, 1000, 40, (2048,)),                                                            # the drivers' 2 x 14 bits
    (11, 4, 1000, 2100, (2048,)),                                                          # a batch past the screen's 2048-row entry
])
def test_wide_coarse_stage_equals_the_oracle(nbits, dc, grid, nq, probes):
    g, ox, xq = _pair(nbits, dc, grid, nq, 77 * nbits + dc + grid)
    kc = 1 << nbits
    for nprobe in probes:
        assert nprobe <= kc * kc
        cd, keys = g.coarse_search(xq, nprobe)
        cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
        assert np.array_equal(keys, keyso), (nbits, dc, grid, nprobe, int((keys != keyso).any(axis=1).sum()))
        assert np.array_equal(bits(cd), bits(cdo)), (nbits, dc, grid, nprobe)
    g.close()


def test_limits():
    g, ox, xq = _pair(6, 8, 1000, 16, 5)
    with pytest.raises(Exception):
        g.coarse_search(xq, 4097)            # VLQ_MAX_IMI_NPROBE + 1
    g.close()
    rng = np.random.default_rng(0)
    f = vlq.GpuIVFPQ(16, 4096, 2, 8)                       # flat quantizer: the reference GPU class's 1024 stays
    f.set_coarse_centroids(rng.random((4096, 16), dtype=np.float32))
    f.set_pq_centroids(rng.random((2, 256, 8), dtype=np.float32))
    with pytest.raises(Exception):
        f.coarse_search(rng.random((4, 16), dtype=np.float32), 1025)
    f.close()


@pytest.mark.parametrize("nbits,dc,Mpq,nb,nprobe,k", [
    (6, 8, 16, 60000, 2048, 10),          # half of the 4096 cells, 16-byte codes, two runs
    (6, 8, 8, 60000, 4096, 100),          # every cell, 8-byte codes (scanm), four runs
    (11, 8, 16, 200000, 2048, 128),       # 2^22 cells: the short-list kernel
    (6, 8, 16, 60000, 1500, 10),          # a ragged second run
])
def test_whole_search_with_more_than_1024_probes(nbits, dc, Mpq, nb, nprobe, k):
    """IndexIVFPQ::search (IndexIVFPQ.cpp:1063-1081) with a multi-index quantizer and the drivers' probe counts: the device
    coarse stage, then runs of 1024 probes scanned and joined -- against the oracle's one long scan."""
    rng = np.random.default_rng(nbits * 100 + Mpq)
    kc, d = 1 << nbits, 2 * dc
    nlist = kc * kc
    cent = rng.random((40, d), dtype=np.float32)
    xb = (cent[rng.integers(0, 40, nb)] + 0.05 * rng.standard_normal((nb, d))).astype(np.float32)
    xq = (cent[rng.integers(0, 40, 150)] + 0.05 * rng.standard_normal((150, d))).astype(np.float32)
    imi = np.stack([xb[rng.choice(nb, kc, replace=False), :dc], xb[rng.choice(nb, kc, replace=False), dc:]]).astype(np.float32)
    pq = (0.05 * rng.standard_normal((Mpq, 256, d // Mpq))).astype(np.float32)
    g = vlq.GpuIVFPQ(d, nlist, Mpq, 8)
    g.set_imi_centroids(nbits, imi)
    g.set_pq_centroids(pq)
    g.set_search_options(by_residual=True, use_precomputed_table=1)
    g.add(xb)
    ox = OracleIndex(d, nlist, Mpq, 8, None, pq, imi_centroids=imi, imi_nbits=nbits, by_residual=1, use_precomputed_table=2)
    ox.add(xb, canonical=True)
    D, I = g.search(xq, nprobe, k)
    Do, Io, keyso, cdo = ox.search(xq, nprobe, k, canonical=True, return_coarse=True)
    assert_same_topk(D, I, Do, Io, "whole search, nprobe %d" % nprobe)
    nq_, ncode = g.stats(reset=True)
    assert nq_ == len(xq)                                   # a query counts once, however its probes were cut
    # the seam with the same probe list: one call, runs inside
    Ds, Is = g.search_preassigned(xq, keyso, cdo, k)
    assert_same_topk(Ds, Is, Do, Io, "preassigned, nprobe %d" % nprobe)
    g.set_search_options(by_residual=True, use_precomputed_table=1, max_codes=1000)
    with pytest.raises(Exception):                          # max_codes would apply per run, not to the whole probe list
        g.search(xq, nprobe, k)
    g.close()


@pytest.mark.parametrize("nbits,d,Mpq,nb,nprobe,k", [
    (11, 16, 8, 150000, 64, 10),          # the multi-index drivers' 8-byte codes on sparse lists
    (11, 16, 8, 150000, 300, 100),
    (11, 16, 4, 150000, 64, 10),
    (11, 96, 12, 150000, 64, 10),         # Deep1B's dimension: 8-dimensional sub-vectors, 48-dimensional halves
    (11, 96, 24, 150000, 64, 128),
    (11, 96, 32, 150000, 64, 10),
    (11, 128, 64, 150000, 32, 300),
    (0, 32, 8, 100000, 48, 10),           # flat quantizer, 65 536 lists: the same kernel without table type 2
])
def test_sparse_lists_of_the_other_code_sizes(nbits, d, Mpq, nb, nprobe, k):
    """A few codes per list with a code size other than 16 bytes (tests/sift1b_imi_pq.cpp and tests/deep1b_imi_pq.cpp ship 8-byte
    codes on 2^28 lists): scanm_short_kernel -- no table per probe, every lane fetches the entries its code addresses -- against
    the oracle bit for bit, and against the generic kernel that served these indexes until round 6."""
    rng = np.random.default_rng(nbits * 1000 + d + Mpq)
    cent = rng.random((40, d), dtype=np.float32)
    xb = (cent[rng.integers(0, 40, nb)] + 0.05 * rng.standard_normal((nb, d))).astype(np.float32)
    xq = (cent[rng.integers(0, 40, 120)] + 0.05 * rng.standard_normal((120, d))).astype(np.float32)
    pq = (0.05 * rng.standard_normal((Mpq, 256, d // Mpq))).astype(np.float32)
    if nbits:
        kc, dc = 1 << nbits, d // 2
        nlist = kc * kc
        imi = np.stack([xb[rng.choice(nb, kc, replace=False), :dc], xb[rng.choice(nb, kc, replace=False), dc:]]).astype(np.float32)
        g = vlq.GpuIVFPQ(d, nlist, Mpq, 8)
        g.set_imi_centroids(nbits, imi)
        ox = OracleIndex(d, nlist, Mpq, 8, None, pq, imi_centroids=imi, imi_nbits=nbits, by_residual=1, use_precomputed_table=2)
    else:
        nlist = 65536
        coarse = (cent[rng.integers(0, 40, nlist)] + 0.05 * rng.standard_normal((nlist, d))).astype(np.float32)
        g = vlq.GpuIVFPQ(d, nlist, Mpq, 8)
        g.set_coarse_centroids(coarse)
        ox = OracleIndex(d, nlist, Mpq, 8, coarse, pq, by_residual=1, use_precomputed_table=1)
    g.set_pq_centroids(pq)
    g.set_search_options(by_residual=True, use_precomputed_table=1)
    g.add(xb)
    ox.add(xb, canonical=True)
    D, I = g.search(xq, nprobe, k)
    assert "scanm_short_kernel<%d>" % Mpq in g.last_scan_info()
    Do, Io, keyso, cdo = ox.search(xq, nprobe, k, canonical=True, return_coarse=True)
    assert_same_topk(D, I, Do, Io, "sparse lists, %d-byte codes" % Mpq)
    _n, ncode = g.stats(reset=True)
    assert ncode == ox.last_ncode
    Dp, Ip = g.search_preassigned(xq, keyso, cdo, k, store_pairs=True)
    Dpo, Ipo = ox.search_preassigned(xq, keyso, cdo, k, store_pairs=True, canonical=True)
    assert_same_topk(Dp, Ip, Dpo, Ipo, "sparse lists, store_pairs")
    g.close()
