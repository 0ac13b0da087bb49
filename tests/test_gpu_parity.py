"""GPU parity (run with -m gpu on an MI355X): the HIP path, called through the
C ABI, against the oracle restatement and the committed reference fixtures."""
import numpy as np
import pytest

import vector_line_quantization_amd as vlq
from oracle import pyoracle
from util import CASE_NAMES, Case, assert_same_topk, bits, label_agreement, sha, tie_canonical

pytestmark = pytest.mark.gpu


def gpu_index(case, with_lists=True):
    g = vlq.GpuIVFPQ(case.d, case.nlist, case.M, case.nbits)
    if case.imi_nbits:
        g.set_imi_centroids(case.imi_nbits, case["imi_centroids"])
    else:
        g.set_coarse_centroids(case["coarse_centroids"])
    g.set_pq_centroids(case["pq_centroids"])
    g.set_search_options(by_residual=case.by_residual, use_precomputed_table=min(case.mode, 1),
                         max_codes=case.max_codes)
    if with_lists:
        g.set_lists(case["codes"], case["ids"], case["list_offsets"])
    return g


@pytest.fixture(scope="module", params=CASE_NAMES)
def case(request):
    return Case(request.param)


def test_query_tables_bit_exact(case):
    g = gpu_index(case)
    n = min(8, case.nq)
    ip = g.query_tables(case.xq[:n], inner_product=True)
    dt = g.query_tables(case.xq[:n], inner_product=False)
    assert np.array_equal(bits(ip[:2]), bits(case["ip_table_head"]))
    assert np.array_equal(sha(ip), case["ip_table_sha256"])
    assert np.array_equal(bits(dt[:2]), bits(case["dis_table_head"]))
    assert np.array_equal(sha(dt), case["dis_table_sha256"])


def test_precomputed_table_bit_exact(case):
    if case.mode not in (1, 2):
        pytest.skip("no precomputed table in this mode")
    g = gpu_index(case)
    t = g.precomputed_table(rows=(1 << case.imi_nbits) if case.imi_nbits else None)
    assert np.array_equal(bits(t[:3]), bits(case["precomputed_table_head"]))
    assert np.array_equal(sha(t), case["precomputed_table_sha256"])


def test_scan_seam_vs_reference_and_oracle(case):
    """search_knn_with_key with the reference's own (keys, coarse_dis)."""
    g = gpu_index(case)
    D, I = g.search_preassigned(case.xq, case["keys"], case["coarse_dis"], case.k)
    # reference: distances bit-exact, labels modulo exact-distance ties
    assert_same_topk(D, I, case["D"], case["I"], case.name)
    # oracle in canonical (distance, scan position) order: everything identical
    ox = case.oracle_index()
    Do, Io = ox.search_preassigned(case.xq, case["keys"], case["coarse_dis"], case.k, canonical=True)
    assert np.array_equal(bits(D), bits(Do))
    assert np.array_equal(I, Io)
    nq, ncode = g.stats(reset=True)
    assert ncode == int(case["ncode"][0]) and nq == case.nq
    Dp, Ip = g.search_preassigned(case.xq, case["keys"], case["coarse_dis"], case.k, store_pairs=True)
    assert_same_topk(Dp, Ip, case["D_pairs"], case["I_pairs"], case.name + " pairs")


def test_coarse_bit_exact_vs_oracle(case):
    g = gpu_index(case)
    ox = case.oracle_index()
    cd, keys = g.coarse_search(case.xq, case.nprobe)
    cdo, keyso = ox.coarse_search(case.xq, case.nprobe, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo))
    assert np.array_equal(keys, keyso)
    if case.imi_nbits and case.d // 2 < 16:   # SSE tables + MinSumK replay: pinned to the reference itself
        assert np.array_equal(keys, case["keys"]) and np.array_equal(bits(cd), bits(case["coarse_dis"]))
    # and against the reference (BLAS summation order unpinned): to rounding
    same = keys == case["keys"]
    assert same.mean() >= 0.999
    ref = case["coarse_dis"]
    rel = np.abs(cd[same] - ref[same]) / np.maximum(np.abs(ref[same]), 1e-20)
    assert rel.max() <= 1e-5


def test_coarse_ties_duplicate_centroids():
    """Many centroids with bit-identical distance at the nprobe boundary: the reference's heap
    keeps the lowest columns (strict `<`, columns visited in order, utils.cpp:876-893); the wave
    select must too, whatever the order its lanes see the columns in."""
    from oracle.pyoracle import OracleIndex
    rng = np.random.default_rng(77)
    d, nlist, M, nprobe = 32, 512, 8, 32
    base = rng.random((8, d)).astype(np.float32)
    cent = base[rng.integers(0, 8, nlist)]                 # 8 distinct rows, ~64 copies each
    pq = rng.random((M, 256, d // M)).astype(np.float32)
    xq = (base[rng.integers(0, 8, 64)] + 0.01 * rng.standard_normal((64, d))).astype(np.float32)
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(cent)
    g.set_pq_centroids(pq)
    ox = OracleIndex(d, nlist, M, 8, cent, pq)
    for nq in (64, 7):                                     # GEMM path and the < 20 queries path
        cd, keys = g.coarse_search(xq[:nq], nprobe)
        cdo, keyso = ox.coarse_search(xq[:nq], nprobe, canonical=True)
        assert np.array_equal(bits(cd), bits(cdo))
        assert np.array_equal(keys, keyso)
        # the lowest columns of the nearest distinct row: 0 .. 31 in column order of that row
        for i in range(nq):
            rows = np.flatnonzero((cent == cent[keys[i, 0]]).all(axis=1))
            assert np.array_equal(keys[i], rows[:nprobe])


@pytest.mark.parametrize("nlist,nprobe", [(256, 1), (1000, 17), (2048, 64), (2052, 33), (5000, 32), (8192, 64), (8196, 8)])
def test_coarse_select_row_sizes(nlist, nprobe):
    """Every row-size class of the register-resident coarse select (and the streaming kernel just
    outside its range) against the oracle's (distance, column) order, bit for bit."""
    from oracle.pyoracle import OracleIndex
    rng = np.random.default_rng(nlist + nprobe)
    d, M = 16, 4
    cent = rng.random((nlist, d)).astype(np.float32)
    cent[nlist // 2:nlist // 2 + 40] = cent[3]              # a run of exact ties
    pq = rng.random((M, 256, d // M)).astype(np.float32)
    xq = rng.random((50, d)).astype(np.float32)
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(cent)
    g.set_pq_centroids(pq)
    ox = OracleIndex(d, nlist, M, 8, cent, pq)
    cd, keys = g.coarse_search(xq, nprobe)
    cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo))
    assert np.array_equal(keys, keyso)


@pytest.mark.parametrize("nlist,d,nprobe", [(16384, 16, 64), (16384, 32, 10), (16448, 8, 1), (32768, 8, 100),
                                            (65536, 4, 256), (131072, 4, 300), (131072, 8, 512)])
def test_wide_rows_two_level_select(nlist, d, nprobe):
    """nlist > 8192, a multiple of 64, at least 4 * nprobe column tiles: the distance kernel also writes
    per-tile minima and the select reads only the tiles that can hold one of the nprobe nearest
    centroids.  Same (distance, column) order as the oracle, with runs of exact ties inside and across
    tiles, and the same leading entries as a selection that is too wide for the two-level path."""
    from oracle.pyoracle import OracleIndex
    rng = np.random.default_rng(nlist + nprobe)
    M = 4
    cent = rng.random((nlist, d)).astype(np.float32)
    cent[nlist // 2:nlist // 2 + 100] = cent[3]             # ties inside a tile and across neighbours
    cent[rng.integers(0, nlist, 500)] = cent[7]              # ties scattered over many tiles
    pq = rng.random((M, 256, d // M)).astype(np.float32)
    xq = rng.random((70, d)).astype(np.float32)
    xq[:8] = cent[3] + 0.001 * rng.standard_normal((8, d)).astype(np.float32)
    xq[8:16] = cent[7]
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(cent)
    g.set_pq_centroids(pq)
    ox = OracleIndex(d, nlist, M, 8, cent, pq)
    cd, keys = g.coarse_search(xq, nprobe)
    cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
    assert np.array_equal(bits(cd), bits(cdo))
    assert np.array_equal(keys, keyso)
    wide = min(1024, nlist // 256 + 1 + nprobe)              # fewer than 4 * nprobe tiles: one-level select
    if wide > nprobe:
        cd2, keys2 = g.coarse_search(xq, wide)
        assert np.array_equal(bits(cd2[:, :nprobe]), bits(cd)) and np.array_equal(keys2[:, :nprobe], keys)


# 1250 / 2500: a 10 000-query batch over 8 / 4 GPUs; 1025 ... 1500, 2100: batches whose last round of workgroups is thin -- its
# queries are split into parts (api.hip: tail_r / tail_p), the whole queries in front of them are not
@pytest.mark.parametrize("nq", [1, 3, 16, 100, 500, 1025, 1031, 1100, 1250, 1500, 2100, 2500, 3000])
def test_small_batches_split_scan(nq):
    """Serving-size batches: a query's probes are split over up to 8 workgroups and the partial rows
    merged -- same distances, same labels, same tie order as the unsplit scan and the oracle."""
    case = Case("c1_small")
    g = gpu_index(case)
    ox = case.oracle_index()
    xq = np.concatenate([case.xq] * (1 + nq // case.xq.shape[0]))[:nq]
    for nprobe, k in ((64, 10), (8, 1), (33, 100), (64, 256)):
        D, I = g.search(xq, nprobe, k)
        Do, Io = ox.search(xq, nprobe, k, canonical=True)
        assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io), (nprobe, k)


def test_coarse_small_batch_matches_reference(case):
    """< 20 queries: the reference takes the SSE path (no BLAS) -> bit-exact."""
    if case.n_small == 0:
        pytest.skip("no small batch in this fixture")
    g = gpu_index(case)
    cd, keys = g.coarse_search(case.xq[:case.n_small], case.nprobe)
    assert_same_topk(cd, keys, case["small_coarse_dis"], case["small_keys"], case.name)
    D, I = g.search(case.xq[:case.n_small], case.nprobe, case.k)
    assert_same_topk(D, I, case["small_D"], case["small_I"], case.name)


def test_full_search_vs_oracle_and_reference(case):
    g = gpu_index(case)
    ox = case.oracle_index()
    D, I = g.search(case.xq, case.nprobe, case.k)
    Do, Io = ox.search(case.xq, case.nprobe, case.k, canonical=True)
    assert np.array_equal(bits(D), bits(Do))
    assert np.array_equal(I, Io)
    Dr, Ir = case["D"], tie_canonical(case["D"], case["I"])
    same = tie_canonical(D, I) == Ir
    assert label_agreement(D, I, case["D"], case["I"]) >= 0.99
    m = same & (Ir >= 0)
    rel = np.abs(D[m] - Dr[m]) / np.maximum(np.abs(Dr[m]), 1e-20)
    assert rel.max() <= 1e-4      # north-star tolerance on distances


def test_encode_and_add(case):
    g = gpu_index(case, with_lists=False)
    ox = case.oracle_index(with_lists=False)
    assign, codes = g.encode(case.xb)
    ao, co = ox.encode(case.xb, canonical=True)
    assert np.array_equal(assign, ao)
    assert np.array_equal(codes, co)
    g.add(case.xb, case.xids)
    ox.add(case.xb, case.xids, canonical=True)
    for i in range(case.nlist):
        c, ids = g.get_list(i)
        o0, o1 = ox.list_offsets[i], ox.list_offsets[i + 1]
        assert np.array_equal(ids, ox.ids[o0:o1])
        assert np.array_equal(c, ox.codes[o0:o1])
    agree = assign == case["xb_assign"]
    assert agree.mean() >= 0.999
    if agree.all():   # then the lists equal the reference's byte for byte
        off = case["list_offsets"]
        for i in range(case.nlist):
            c, ids = g.get_list(i)
            assert np.array_equal(ids, case["ids"][off[i]:off[i + 1]])
            assert np.array_equal(c, case["codes"][off[i]:off[i + 1]])


@pytest.mark.parametrize("name", ["c1_small", "deep_like_dsub6", "imi_sse_tables"])
def test_add_in_batches_appends_on_device(name):
    """add() in uneven batches (device-side append: in-place when the slack suffices, relayout
    with 25 % slack otherwise) == one add() == the reference's lists, incl. order inside a list
    (IndexIVFPQ.cpp:236-248); searches before, between and after see exactly the stored vectors."""
    case = Case(name)
    g1 = gpu_index(case, with_lists=False)
    g1.add(case.xb, case.xids)
    g = gpu_index(case, with_lists=False)
    cuts = [0, 1, 2, 700, 701, 1900, len(case.xb)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        g.add(case.xb[a:b], None if case.xids is None else case.xids[a:b])
        assert g.ntotal == b
        if b == 701:       # a search in the middle of the build only sees the first 701 vectors
            D, I = g.search(case.xq[:8], case.nprobe, case.k)
            seen = set(range(701)) if case.xids is None else set(case.xids[:701].tolist())
            assert set(I[I >= 0].tolist()) <= seen
    for i in range(case.nlist):
        c, ids = g.get_list(i)
        c1, ids1 = g1.get_list(i)
        assert np.array_equal(ids, ids1) and np.array_equal(c, c1)
    D, I = g.search(case.xq, case.nprobe, case.k)
    D1, I1 = g1.search(case.xq, case.nprobe, case.k)
    assert np.array_equal(bits(D), bits(D1)) and np.array_equal(I, I1)
    # reserveMemory / reclaimMemory: layout changes only
    g4 = gpu_index(case, with_lists=False)
    g4.reserve_memory(2 * len(case.xb))
    g4.add(case.xb[:1000], None if case.xids is None else case.xids[:1000])
    g4.add(case.xb[1000:], None if case.xids is None else case.xids[1000:])
    assert g4.reclaim_memory() > 0 and g4.reclaim_memory() == 0
    assert g.reclaim_memory() > 0
    for gx in (g4, g):
        for i in range(case.nlist):
            c, ids = gx.get_list(i)
            c1, ids1 = g1.get_list(i)
            assert np.array_equal(ids, ids1) and np.array_equal(c, c1)
        Dx, Ix = gx.search(case.xq, case.nprobe, case.k)
        assert np.array_equal(bits(Dx), bits(D1)) and np.array_equal(Ix, I1)
    # packed lists loaded with set_lists and grown by add afterwards
    g2 = gpu_index(case, with_lists=False)
    g2.add(case.xb[:1900], None if case.xids is None else case.xids[:1900])
    off = np.zeros(case.nlist + 1, np.int64)
    cs, ids_ = [], []
    for i in range(case.nlist):
        c, ids = g2.get_list(i)
        cs.append(c); ids_.append(ids); off[i + 1] = off[i] + len(ids)
    g3 = gpu_index(case, with_lists=False)
    g3.set_lists(np.concatenate(cs), np.concatenate(ids_), off)
    g3.add(case.xb[1900:], None if case.xids is None else case.xids[1900:])
    for i in range(case.nlist):
        c, ids = g3.get_list(i)
        c1, ids1 = g1.get_list(i)
        assert np.array_equal(ids, ids1) and np.array_equal(c, c1)


@pytest.mark.parametrize("k", [1, 64, 65, 100, 128, 129, 256, 257, 512, 513, 1024])
def test_large_k_and_k_boundaries(k):
    """k at the edges of the per-lane key counts of the wave select (1/4/16)."""
    case = Case("deep_like_dsub6")
    g = gpu_index(case)
    ox = case.oracle_index()
    D, I = g.search_preassigned(case.xq, case["keys"], case["coarse_dis"], k)
    Do, Io = ox.search_preassigned(case.xq, case["keys"], case["coarse_dis"], k, canonical=True)
    assert np.array_equal(bits(D), bits(Do))
    assert np.array_equal(I, Io)


@pytest.mark.parametrize("nprobe", [1, 37, 64, 65, 300])
def test_nprobe_edges(nprobe):
    """nprobe above nlist pads the coarse result with -1 keys which the scan skips."""
    case = Case("c1_small")
    g = gpu_index(case)
    ox = case.oracle_index()
    D, I = g.search(case.xq, nprobe, 10)
    Do, Io = ox.search(case.xq, nprobe, 10, canonical=True)
    assert np.array_equal(bits(D), bits(Do))
    assert np.array_equal(I, Io)


def test_empty_and_error_paths():
    case = Case("tiny_padding")
    g = gpu_index(case, with_lists=False)
    D, I = g.search(case.xq, 4, 5)                      # empty index: all padding
    assert (I == -1).all() and (D == np.finfo(np.float32).max).all()
    D, I = g.search(case.xq[:0], 4, 5)                  # empty batch
    assert D.shape == (0, 5)
    with pytest.raises(vlq.VlqError):
        g.search(case.xq, 0, 5)
    with pytest.raises(vlq.VlqError):
        g.search(case.xq, 4, 2000)
    bad = np.full((case.nq, 2), case.nlist + 3, np.int64)
    with pytest.raises(vlq.VlqError):                    # reference: "Invalid key" + throw (IndexIVFPQ.cpp:1008-1011)
        g.search_preassigned(case.xq, bad, np.zeros((case.nq, 2), np.float32), 3)
    g.stats()                                            # ... reported once, then cleared
    # device outputs: the call stays asynchronous, the error surfaces at the next stats()
    import torch
    Dd = torch.empty((case.nq, 3), dtype=torch.float32, device="cuda")
    Id = torch.empty((case.nq, 3), dtype=torch.int64, device="cuda")
    g.search_preassigned(torch.from_numpy(case.xq).cuda(), torch.from_numpy(bad).cuda(),
                         torch.zeros((case.nq, 2), dtype=torch.float32, device="cuda"), 3, D=Dd, I=Id)
    with pytest.raises(vlq.VlqError):
        g.stats()
    with pytest.raises(ValueError):                      # outputs are written in place: no silent temporaries
        g.search(case.xq, 4, 3, D=np.empty((case.nq, 6), np.float32)[:, ::2], I=np.empty((case.nq, 3), np.int64))


def test_negative_user_ids_survive_every_path():
    """add_with_ids accepts any int64 (IndexIVFPQ.cpp:236-248): negative ids must come back from the
    single-workgroup scan, the split scan + merge (small batches) and vlq_merge_topk alike."""
    case = Case("c1_small")
    g = gpu_index(case, with_lists=False)
    neg = -(np.arange(case.xb.shape[0], dtype=np.int64) + 5)
    g.add(case.xb, neg)
    ox = case.oracle_index(with_lists=False)
    ox.add(case.xb, neg, canonical=True)
    for nq in (2, case.nq):                              # 2 queries: probes split over workgroups + merge
        D, I = g.search(case.xq[:nq], case.nprobe, case.k)
        Do, Io = ox.search(case.xq[:nq], case.nprobe, case.k, canonical=True)
        assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)
        assert (I[D < np.finfo(np.float32).max] < 0).all()


def test_device_resident_buffers():
    """Inputs and outputs already in HBM (torch tensors as plain device memory)."""
    import torch
    case = Case("c1_small")
    g = gpu_index(case)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    xq = torch.from_numpy(case.xq).cuda()
    D, I = g.search(xq, case.nprobe, case.k)
    torch.cuda.synchronize()
    ox = case.oracle_index()
    Do, Io = ox.search(case.xq, case.nprobe, case.k, canonical=True)
    assert np.array_equal(bits(D.cpu().numpy()), bits(Do))
    assert np.array_equal(I.cpu().numpy(), Io)


def test_stage_timer_modes():
    """vlq_ivfpq_profile: 1 = every stage of every call, 2 = the scan launch of every call, 3 = the scan launch of every
    4th call starting with the next one (what bench.py's timed region uses); results do not depend on the mode."""
    case = Case("c1_small")
    g = gpu_index(case)
    D0, I0 = g.search(case.xq, case.nprobe, case.k)
    for mode, want_calls, stages in ((1, 8, True), (2, 8, False), (3, 2, False), (0, 0, False)):
        g.profile(mode)
        g.profile_read(reset=True)
        for _ in range(8):
            D, I = g.search(case.xq, case.nprobe, case.k)
            assert np.array_equal(bits(D), bits(D0)) and np.array_equal(I, I0)
        p = g.profile_read(reset=True)
        assert p["scan_calls"] == want_calls, (mode, p)
        assert (p["scan_ms"] > 0) == (want_calls > 0)
        assert (p["coarse_ms"] > 0) == stages
    g.profile(False)


@pytest.mark.parametrize("nq", [100, 4096, 5000])
def test_host_buffers_pageable_and_page_locked(nq):
    """The reference drivers' calling convention: x / D / I in host memory.  4 096 queries and more: the batch is copied
    in four chunks, each beside the coarse GEMM of the chunk before; page-locked outputs
    (GpuResources::getPinnedMemory) are written by the scan kernel itself.  Same rows as the device-resident call, which the oracle checks above."""
    import torch
    case = Case("c1_small")
    g = gpu_index(case)
    xq = np.concatenate([case.xq] * (1 + nq // case.xq.shape[0]))[:nq]
    xd = torch.from_numpy(xq).cuda()
    Dd, Id = g.search(xd, case.nprobe, case.k)
    torch.cuda.synchronize()
    Dd, Id = Dd.cpu().numpy(), Id.cpu().numpy()
    ox = case.oracle_index()
    Do, Io = ox.search(xq[:64], case.nprobe, case.k, canonical=True)
    assert np.array_equal(bits(Dd[:64]), bits(Do)) and np.array_equal(Id[:64], Io)
    D, I = g.search(xq, case.nprobe, case.k)                         # pageable numpy
    assert np.array_equal(bits(D), bits(Dd)) and np.array_equal(I, Id)
    xp = torch.from_numpy(xq).pin_memory()
    Dp = torch.full((nq, case.k), -1.0, dtype=torch.float32).pin_memory()
    Ip = torch.full((nq, case.k), -7, dtype=torch.int64).pin_memory()
    for _ in range(2):                                               # second call: the staging buffer is reused
        g.search(xp.numpy(), case.nprobe, case.k, D=Dp.numpy(), I=Ip.numpy())
        assert np.array_equal(bits(Dp.numpy()), bits(Dd)) and np.array_equal(Ip.numpy(), Id)
    # mixed: page-locked distances, pageable labels
    I2 = np.empty((nq, case.k), np.int64)
    Dp.fill_(-1.0)
    g.search(xq, case.nprobe, case.k, D=Dp.numpy(), I=I2)
    assert np.array_equal(bits(Dp.numpy()), bits(Dd)) and np.array_equal(I2, Id)


@pytest.mark.parametrize("k,nparts", [(10, 2), (100, 8), (300, 3)])
def test_merge_topk(k, nparts):
    """vlq_merge_topk (list-sharded mode): the k smallest over the shards' sorted rows,
    ties to the lower shard; -1 / FLT_MAX padding never wins."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(k)
    nq = 37
    D = np.sort(rng.integers(0, 50, (nparts, nq, k)).astype(np.float32), axis=2)   # many ties
    I = rng.integers(0, 10 ** 9, (nparts, nq, k)).astype(np.int64)
    D[1, :, k // 2:] = np.finfo(np.float32).max
    I[1, :, k // 2:] = -1
    I[0, :, 0] = -7 - np.arange(nq)             # negative ids are data, not padding
    Dd, Id = torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda()
    Do = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    Io = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    from vector_line_quantization_amd._lib import check, lib
    check(lib().vlq_merge_topk(C.c_int(0), C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_int64(nq),
                               C.c_int(k), C.c_int(nparts), C.c_void_p(Dd.data_ptr()), C.c_void_p(Id.data_ptr()),
                               C.c_void_p(Do.data_ptr()), C.c_void_p(Io.data_ptr())))
    torch.cuda.synchronize()
    allD = np.concatenate(list(D), axis=1)
    allI = np.concatenate(list(I), axis=1)
    order = np.argsort(np.where(allD == np.finfo(np.float32).max, np.inf, allD), axis=1, kind="stable")[:, :k]
    assert np.array_equal(Do.cpu().numpy(), np.take_along_axis(allD, order, 1))
    assert np.array_equal(Io.cpu().numpy(), np.take_along_axis(allI, order, 1))


@pytest.mark.parametrize("nbits,dc,nprobe", [(4, 8, 1), (4, 8, 2), (5, 8, 63), (5, 32, 64), (6, 8, 100), (6, 32, 127),
                                             (7, 8, 128), (7, 16, 200), (8, 8, 255), (8, 8, 256), (5, 8, 700)])
def test_multi_index_walk_sizes_and_ties(nbits, dc, nprobe):
    """MinSumK (IndexPQ.cpp:690-778) replayed on the device (heap in LDS, or in global memory when it
    does not fit): keys and path-dependent float sums equal to the oracle's replay bit for bit over the
    heap sizes either side of the switch, with duplicate sub-centroids (equal sums: the heap's own tie
    order) and both table sources (SSE for 8-dim halves, the distance kernel for wider ones)."""
    from oracle.pyoracle import OracleIndex
    rng = np.random.default_rng(nbits * 1000 + nprobe)
    kc, d, M = 1 << nbits, 2 * dc, 4
    imi = rng.random((2, kc, dc)).astype(np.float32)
    imi[0, kc // 2:kc // 2 + 3] = imi[0, 1]                   # exact ties between cells
    imi[1, 5:7] = imi[1, 0]
    pq = rng.random((M, 256, d // M)).astype(np.float32)
    xq = rng.random((37, d)).astype(np.float32)
    xq[:4, :dc] = imi[0, 1]
    g = vlq.GpuIVFPQ(d, kc * kc, M, 8)
    g.set_imi_centroids(nbits, imi)
    g.set_pq_centroids(pq)
    ox = OracleIndex(d, kc * kc, M, 8, None, pq, imi_centroids=imi, imi_nbits=nbits)
    npb = min(nprobe, kc * kc)
    cd, keys = g.coarse_search(xq, npb)
    cdo, keyso = ox.coarse_search(xq, npb, canonical=True)
    assert np.array_equal(keys, keyso)
    assert np.array_equal(bits(cd), bits(cdo))


@pytest.mark.parametrize("nlist,d", [(256, 16), (4096, 32), (16384, 8), (320, 12), (131072, 4)])
def test_nearest_centroid_without_distance_matrix(nlist, d):
    """nprobe = 1 on >= 20 rows (the assignment of add / encode): the distance kernel emits one
    (distance, column) key per 64-column tile instead of the matrix.  Same distance bits and the lowest
    column among exact ties (duplicate centroids inside a tile, across tiles, across row blocks), as the
    oracle and as the first entry of a wider selection; encode() assigns accordingly."""
    from oracle.pyoracle import OracleIndex
    rng = np.random.default_rng(nlist + d)
    M = 4
    cent = rng.random((nlist, d)).astype(np.float32)
    cent[nlist // 2:nlist // 2 + 70] = cent[3]                # the nearest centroid of some rows, many copies
    cent[rng.integers(0, nlist, 40)] = cent[9]
    pq = rng.random((M, 256, d // M)).astype(np.float32)
    xq = rng.random((300, d)).astype(np.float32)
    xq[:40] = cent[3] + 0.001 * rng.standard_normal((40, d)).astype(np.float32)
    xq[40:60] = cent[9]
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(cent)
    g.set_pq_centroids(pq)
    ox = OracleIndex(d, nlist, M, 8, cent, pq)
    cd, keys = g.coarse_search(xq, 1)
    cdo, keyso = ox.coarse_search(xq, 1, canonical=True)
    assert np.array_equal(keys, keyso)
    assert np.array_equal(bits(cd), bits(cdo))
    cd3, keys3 = g.coarse_search(xq, 3)
    assert np.array_equal(keys3[:, :1], keys) and np.array_equal(bits(cd3[:, :1]), bits(cd))
    assign, _codes = g.encode(xq)
    assert np.array_equal(assign, keys[:, 0])


@pytest.mark.parametrize("k", [200, 300, 1000])
def test_multi_index_with_workgroup_selection(k):
    """Inverted multi-index + table type 2 through the one-selection-per-workgroup kernel (k > 128,
    scan16_bigk_kernel<.., IMI>): lists long enough for the per-probe table kernel, ties from duplicated
    vectors, against the oracle."""
    from oracle.pyoracle import OracleIndex
    rng = np.random.default_rng(k)
    nbits, d, M = 3, 128, 16
    kc, dc = 1 << nbits, d // 2
    imi = rng.random((2, kc, dc), dtype=np.float32)
    pq = ((rng.random((M, 256, d // M), dtype=np.float32) - 0.5) * 0.4).astype(np.float32)
    xb = rng.random((6000, d), dtype=np.float32)
    xb[3000:3400] = xb[100:500]                       # exact duplicates: equal distances
    xq = rng.random((150, d), dtype=np.float32)
    ox = OracleIndex(d, kc * kc, M, 8, None, pq, imi_centroids=imi, imi_nbits=nbits)
    ox.add(xb, canonical=True)
    g = vlq.GpuIVFPQ(d, kc * kc, M, 8)
    g.set_imi_centroids(nbits, imi)
    g.set_pq_centroids(pq)
    g.set_lists(ox.codes, ox.ids, ox.list_offsets)
    D, I = g.search(xq, 16, k)
    Do, Io = ox.search(xq, 16, k, canonical=True)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)


@pytest.mark.parametrize("nlist,d,nprobe", [(4096, 128, 32), (8192, 64, 16), (16384, 96, 64)])
def test_filtered_coarse_stage_is_exact(monkeypatch, nlist, d, nprobe):
    """The matrix-free coarse stage (VLQ_COARSE_FILTER=1: sampled bound, filtered GEMM epilogue, select
    over the kept keys; off by default because it is slower) returns the oracle's keys and distances bit
    for bit -- including rows whose tiles overflow their 16-key groups (a run of 40 duplicated centroids
    next to some queries) and are redone as fmaf chains."""
    from oracle.pyoracle import OracleIndex
    monkeypatch.setenv("VLQ_COARSE_FILTER", "1")
    rng = np.random.default_rng(nlist + nprobe)
    coarse = rng.random((nlist, d), dtype=np.float32)
    coarse[200:240] = coarse[7]                       # 40 equal columns inside one tile: more than a group holds
    pq = rng.random((4, 256, d // 4), dtype=np.float32)
    xq = rng.random((600, d), dtype=np.float32)
    xq[:20] = coarse[7] + 0.001 * rng.standard_normal((20, d)).astype(np.float32)
    g = vlq.GpuIVFPQ(d, nlist, 4, 8)
    g.set_coarse_centroids(coarse)
    g.set_pq_centroids(pq)
    ox = OracleIndex(d, nlist, 4, 8, coarse, pq)
    cd, keys = g.coarse_search(xq, nprobe)
    cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
    assert np.array_equal(keys, keyso)
    assert np.array_equal(bits(cd), bits(cdo))


def test_encode_preassigned_matches_encode():
    """vlq_ivfpq_encode_preassigned (IndexIVFPQ::encode_multiple with compute_keys = false / add_core_o's
    precomputed_idx): the codes of encode() when given encode()'s lists; a negative list encodes a zero residual
    (IndexIVFPQ.cpp:219-221)."""
    case = Case("c1_small")
    g = gpu_index(case, with_lists=False)
    x = case.xb[:3000]
    assign, codes = g.encode(x)
    assert np.array_equal(g.encode_preassigned(x, assign), codes)
    a2 = assign.copy()
    a2[::7] = (a2[::7] + 1) % case.nlist              # other lists: other residuals
    a2[5] = -1
    c2 = g.encode_preassigned(x, a2)
    same = a2 == assign
    assert np.array_equal(c2[same], codes[same]) and (c2[~same] != codes[~same]).any()
    # a vector encoded for the "wrong" list = that list's own encoding of the vector: the oracle's encode of
    # x - c[a2] + c[a'] ... is checked directly instead: residual to list a2, PQ argmin per sub-quantizer
    cent, pq = case["coarse_centroids"], case["pq_centroids"].reshape(case.M, 1 << case.nbits, -1)
    for i in (7, 14, 700):
        r = (x[i] - cent[a2[i]]).reshape(case.M, -1)
        ref = [int(np.argmin(((pq[m] - r[m]) ** 2).sum(1))) for m in range(case.M)]
        assert (np.asarray(ref) == c2[i]).mean() >= 0.9          # float32 vs numpy summation order: near-ties may differ
    # key < 0  ==  zero residual  ==  a vector sitting on its centroid
    zero = g.encode_preassigned(cent[3:4].copy(), np.array([3], np.int64))
    assert np.array_equal(c2[5], zero[0])
