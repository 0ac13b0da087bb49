"""GPU: the float16 screen of the coarse stage (csrc/coarse_screen.hip; include/vlq_ivfpq.h vlq_ivfpq_set_coarse_screen)
never shows in a result: batches of 2048 queries and more against the oracle's coarse assignment -- knn_L2sqr's
(distance, column) order (utils.cpp:884, :935-946; Heap.h) -- bit for bit, on every row-size class, with runs of exact
ties, on data that defeat the screen's error bound (every row is then done exactly and the index drops the screen by
itself), with queries outside the half range and with queries that are not numbers."""
import numpy as np
import pytest

import vector_line_quantization_amd as vlq
from oracle.pyoracle import OracleIndex
from util import bits

pytestmark = pytest.mark.gpu

NQ = 2200          # >= 2048: the screened path; 2200 = 17 row blocks of 128 + 24 rows


def make(nlist, d, rng, offset=0.0, spread=1.0):
    M = 4 if d % 4 == 0 and d >= 4 else 1
    cent = (offset + spread * rng.random((nlist, d))).astype(np.float32)
    cent[nlist // 2:nlist // 2 + 40] = cent[3]              # a run of exact ties
    pq = rng.random((M, 256, d // M)).astype(np.float32)
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(cent)
    g.set_pq_centroids(pq)
    return g, OracleIndex(d, nlist, M, 8, cent, pq), cent


@pytest.mark.parametrize("nlist,d,nprobe,decides", [(256, 16, 8, True), (1024, 64, 32, True), (2048, 100, 17, True),
                                                    (4096, 128, 32, True), (4096, 96, 2, True), (8192, 8, 32, True),
                                                    (8192, 32, 64, None), (1024, 32, 64, True), (4096, 64, 128, None), (1024, 32, 100, None),
                                                    # rows wider than 8192 columns: tile minima from the distance kernel, tiled keep kernel
                                                    (16384, 64, 32, None), (65536, 16, 8, None), (131072, 8, 16, None), (16448, 32, 64, None),
                                                    (131072, 16, 128, None), (8256, 32, 128, None),
                                                    # round 6: the matrix-free passes beyond 16 384 lists (several ranges per workgroup,
                                                    # the last one ragged) and beyond 64 probes (the bound by bisection, two result registers)
                                                    (32768, 96, 100, None), (65536, 64, 64, None), (131072, 32, 40, None), (20032, 32, 65, None),
                                                    (16384, 48, 128, None)])
def test_screened_coarse_equals_oracle(nlist, d, nprobe, decides):
    rng = np.random.default_rng(nlist + d + nprobe)
    g, ox, cent = make(nlist, d, rng)
    xq = rng.random((NQ, d)).astype(np.float32)
    xq[:64] = cent[rng.integers(0, nlist, 64)]               # queries ON centroids: distance 0 and, for the tie run, ties at 0
    cd, keys = g.coarse_search(xq, nprobe)
    en, rows, und = g.coarse_screen_state()
    assert rows == NQ                                        # the screen ran ...
    if decides is not None:
        assert (en and und == 0) if decides else und > 0      # ... and decided every row, where the data let it
    cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo))
    assert np.array_equal(keys, keyso)
    g.set_coarse_screen(0)                                    # and the matrix path gives the same (the switch is speed only)
    cd0, keys0 = g.coarse_search(xq, nprobe)
    assert np.array_equal(bits(cd0), bits(cd)) and np.array_equal(keys0, keys)


def test_offset_data_are_centred():
    """Centroids 5 + uniform(0, 1): the norms are 25 times those of the centred data -- the half arithmetic works on q - mu,
    c - mu (mu = the centroids' mean), only the exact stage's own rounding term keeps the uncentred norms."""
    rng = np.random.default_rng(5)
    nlist, d, nprobe = 512, 32, 16
    g, ox, cent = make(nlist, d, rng, offset=5.0, spread=1.0)
    xq = (5.0 + rng.random((NQ, d))).astype(np.float32)
    cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
    cd, keys = g.coarse_search(xq, nprobe)
    assert np.array_equal(bits(cd), bits(cdo)) and np.array_equal(keys, keyso)
    en, rows, und = g.coarse_screen_state()
    assert en and rows == NQ and und * 200 <= NQ


def test_data_that_defeat_the_bound_are_done_exactly_and_drop_the_screen():
    """Centroids and queries 100 + uniform(0, 1) in 16 dimensions: the exact fp32 distances -- differences of numbers near
    320 000 -- are themselves uncertain by more than their spread, the bound (rightly) keeps every column, every row is done
    exactly in full, and the index drops the screen by itself."""
    rng = np.random.default_rng(7)
    nlist, d, nprobe = 1024, 16, 8
    g, ox, cent = make(nlist, d, rng, offset=100.0, spread=1.0)
    xq = (100.0 + rng.random((NQ, d))).astype(np.float32)
    cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
    for it in range(3):
        cd, keys = g.coarse_search(xq, nprobe)
        assert np.array_equal(bits(cd), bits(cdo)) and np.array_equal(keys, keyso), it
    en, rows, und = g.coarse_screen_state()
    assert not en and und * 200 > NQ and rows <= 2 * NQ       # dropped once the first batch's counters reached the host


def test_queries_outside_the_half_range_and_nan_rows():
    rng = np.random.default_rng(6)
    nlist, d, nprobe = 1024, 32, 8
    g, ox, cent = make(nlist, d, rng)
    xq = rng.random((NQ, d)).astype(np.float32)
    xq[100:105] *= 1.0e6                                      # |q_i| * scale > 65504: flagged rows, exact path
    xq[500, 3] = np.float32(3.0e38)                           # squared norm overflows to inf
    cd, keys = g.coarse_search(xq, nprobe)
    ok = np.ones(NQ, bool)
    ok[500] = False
    cdo, keyso = ox.coarse_search(xq[ok], nprobe, canonical=True)
    assert np.array_equal(bits(cd[ok]), bits(cdo)) and np.array_equal(keys[ok], keyso)
    en, rows, und = g.coarse_screen_state()
    assert en and 5 <= und <= 8                               # the flagged rows (and the overflowing one), not the rest;
                                                              # under 0.5 % of the batch: the screen stays
    # rows that are not numbers: whatever the matrix path returns for them, the screened path returns too
    xq[700, 0] = np.nan
    xq[701] = np.inf
    cd1, keys1 = g.coarse_search(xq, nprobe)
    g.set_coarse_screen(0)
    cd0, keys0 = g.coarse_search(xq, nprobe)
    assert np.array_equal(bits(cd1), bits(cd0)) and np.array_equal(keys1, keys0)


@pytest.mark.parametrize("nlist,d,nprobe", [(1024, 32, 8), (4096, 128, 32), (16384, 64, 32), (131072, 16, 100)])
@pytest.mark.parametrize("far", [1.5, 2.0, 3.0, 4.0, 8.0])
def test_outlier_queries_near_the_half_overflow(nlist, d, nprobe, far):
    """Queries at 1.5 ... 8 times the centroid cloud's radius from its centre: their stored half distances reach the top
    of the half range, where a value that rounds to +inf can never pass `w <= T`.  A bound at or above 65504 in the stored
    domain must send the row to the exact path (screen_threshold's tmax): keys and distances equal the oracle's."""
    rng = np.random.default_rng(int(nlist + d + 10 * far))
    g, ox, cent = make(nlist, d, rng)
    mu = cent.mean(0)
    radius = np.sqrt(((cent - mu) ** 2).sum(1)).max()
    dirs = rng.standard_normal((NQ, d)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    scale = (far * (0.8 + 0.4 * rng.random((NQ, 1)))).astype(np.float32)
    xq = (mu + dirs * radius * scale).astype(np.float32)
    xq[:200] = rng.random((200, d)).astype(np.float32)            # ordinary rows among them
    cd, keys = g.coarse_search(xq, nprobe)
    cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo))
    assert np.array_equal(keys, keyso)


@pytest.mark.parametrize("nbits,d,nprobe", [(8, 128, 16), (10, 64, 64)])
def test_multi_index_halves_go_through_the_screen(nbits, d, nprobe):
    """Inverted multi-index: each half's table of 2^nbits sub-centroids is screened like a flat quantizer's; the walk over
    the two sorted lists (MinSumK) sees the same T nearest sub-centroids and distances, so the cells and their sums are
    the oracle's."""
    rng = np.random.default_rng(nbits * 100 + nprobe)
    kc, dc, M = 1 << nbits, d // 2, 16
    imi = rng.random((2, kc, dc), dtype=np.float32)
    imi[0, 7:20] = imi[0, 5]                                  # ties among sub-centroids
    pq = ((rng.random((M, 256, d // M), dtype=np.float32) - 0.5) * 0.4).astype(np.float32)
    xq = rng.random((NQ, d), dtype=np.float32)
    ox = OracleIndex(d, kc * kc, M, 8, None, pq, imi_centroids=imi, imi_nbits=nbits)
    g = vlq.GpuIVFPQ(d, kc * kc, M, 8)
    g.set_imi_centroids(nbits, imi)
    g.set_pq_centroids(pq)
    cd, keys = g.coarse_search(xq, nprobe)
    en, rows, und = g.coarse_screen_state()
    assert rows == 2 * NQ                                     # both halves
    cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo))
    assert np.array_equal(keys, keyso)
    g.set_coarse_screen(0)
    cd0, keys0 = g.coarse_search(xq, nprobe)
    assert np.array_equal(bits(cd0), bits(cd)) and np.array_equal(keys0, keys)


def test_more_than_128_probes_take_the_matrix_path():
    rng = np.random.default_rng(9)
    g, ox, cent = make(1024, 32, rng)
    xq = rng.random((NQ, 32)).astype(np.float32)
    cd, keys = g.coarse_search(xq, 130)
    assert g.coarse_screen_state()[1] == 0                    # no row went through the screen
    cdo, keyso = ox.coarse_search(xq, 130, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo)) and np.array_equal(keys, keyso)


@pytest.mark.parametrize("seed", range(4))
def test_near_ties_at_every_scale_around_the_bound(seed):
    """Clusters of 16 centroids whose mutual distances span 1e-6 .. 1e-1 of their norms -- below, at and above the screen's
    error bound -- with nprobe cutting through clusters: a column the bound wrongly dropped would show as a wrong key."""
    rng = np.random.default_rng(100 + seed)
    nlist, d, nprobe = 1024, 64, 24
    centres = rng.random((nlist // 16, d)).astype(np.float32) * 4 - 1
    eps = 10.0 ** rng.uniform(-6, -1, size=(nlist // 16, 1, 1))
    cent = (centres[:, None, :] * (1 + eps * rng.standard_normal((nlist // 16, 16, d)))).reshape(nlist, d).astype(np.float32)
    pq = rng.random((4, 256, d // 4)).astype(np.float32)
    g = vlq.GpuIVFPQ(d, nlist, 4, 8)
    g.set_coarse_centroids(cent)
    g.set_pq_centroids(pq)
    ox = OracleIndex(d, nlist, 4, 8, cent, pq)
    xq = (centres[rng.integers(0, nlist // 16, NQ)] * (1 + 10.0 ** rng.uniform(-5, -1, size=(NQ, 1)) * rng.standard_normal((NQ, d)))).astype(np.float32)
    cd, keys = g.coarse_search(xq, nprobe)
    assert g.coarse_screen_state()[1] == NQ
    cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo))
    assert np.array_equal(keys, keyso)


@pytest.mark.parametrize("nlist,d", [(256, 16), (4096, 128), (16384, 64), (131072, 8), (1088, 100)])
def test_nearest_centroid_screened(nlist, d):
    """nprobe 1 = the assignment of add / encode: approximate tile minima only, the tiles under the bound exactly -- the
    matrix path's arg-min by (distance, column), ties at distance 0 included."""
    rng = np.random.default_rng(nlist + d)
    g, ox, cent = make(nlist, d, rng)
    xq = rng.random((NQ, d)).astype(np.float32)
    xq[:200] = cent[rng.integers(0, nlist, 200)]             # on centroids; the tie run (make()) among them
    xq[200:240] = cent[3]
    cd, keys = g.coarse_search(xq, 1)
    en, rows, und = g.coarse_screen_state()
    assert rows == NQ
    cdo, keyso = ox.coarse_search(xq, 1, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo))
    assert np.array_equal(keys, keyso)
    g.set_coarse_screen(0)
    cd0, keys0 = g.coarse_search(xq, 1)
    assert np.array_equal(bits(cd0), bits(cd)) and np.array_equal(keys0, keys)


def test_add_goes_through_the_screen_and_fills_the_same_lists():
    rng = np.random.default_rng(11)
    nlist, d, M = 512, 64, 16
    cent = rng.random((nlist, d)).astype(np.float32)
    pq = (0.2 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    xb = (cent[rng.integers(0, nlist, 6000)] + 0.05 * rng.standard_normal((6000, d))).astype(np.float32)
    lists = []
    for screen in (1, 0):
        g = vlq.GpuIVFPQ(d, nlist, M, 8)
        g.set_coarse_centroids(cent)
        g.set_pq_centroids(pq)
        g.set_coarse_screen(screen)
        g.add(xb)
        assert (g.coarse_screen_state()[1] > 0) == bool(screen)
        lists.append([g.get_list(i) for i in range(nlist)])
    for (c1, i1), (c0, i0) in zip(*lists):
        assert np.array_equal(i1, i0) and np.array_equal(c1, c0)
    ox = OracleIndex(d, nlist, M, 8, cent, pq)
    ox.add(xb, None, canonical=True)
    off = ox.list_offsets
    for i in range(nlist):
        assert np.array_equal(lists[0][i][1], ox.ids[off[i]:off[i + 1]])
        assert np.array_equal(lists[0][i][0], ox.codes[off[i]:off[i + 1]])


def test_multi_index_assignment_screened():
    rng = np.random.default_rng(12)
    nbits, d, M = 8, 128, 16
    kc, dc = 1 << nbits, d // 2
    imi = rng.random((2, kc, dc), dtype=np.float32)
    imi[1, 9:30] = imi[1, 4]
    pq = ((rng.random((M, 256, d // M), dtype=np.float32) - 0.5) * 0.4).astype(np.float32)
    xq = rng.random((NQ, d), dtype=np.float32)
    xq[:50, dc:] = imi[1, 4]
    ox = OracleIndex(d, kc * kc, M, 8, None, pq, imi_centroids=imi, imi_nbits=nbits)
    g = vlq.GpuIVFPQ(d, kc * kc, M, 8)
    g.set_imi_centroids(nbits, imi)
    g.set_pq_centroids(pq)
    cd, keys = g.coarse_search(xq, 1)
    assert g.coarse_screen_state()[1] == 2 * NQ
    cdo, keyso = ox.coarse_search(xq, 1, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo)) and np.array_equal(keys, keyso)
