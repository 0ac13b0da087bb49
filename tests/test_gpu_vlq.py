"""GPU: the VLQ (vector and line quantization) path through the C ABI of
include/vlq_line.h against the VLQ oracle.  The oracle itself is PARITY UNPINNED
against the reference binary (CUDA-only path, see oracle/vlq_oracle.cpp) and is held
to independent float64 recomputations in tests/test_vlq_oracle.py; here the HIP path
must reproduce the oracle bit for bit."""
import numpy as np
import pytest

import vector_line_quantization_amd as vlq
from test_vlq_oracle import decode, make_vlq
from util import bits

pytestmark = pytest.mark.gpu


def gpu_from_oracle(v, with_lists=True, graph=True):
    g = vlq.GpuVLQ(v.d, v.nlist, v.M, v.nbits, v.nedge, v.nlambda)
    g.set_coarse_centroids(v.coarse)
    if graph:
        g.set_graph(v.edge_info, v.edge_dist)
    g.set_lambda_codebook(v.lambda_info)
    g.set_pq_centroids(v.pq_centroids)
    if with_lists:
        g.set_lists(v.codes, v.lambdas, v.ids, v.line_off)
    return g


@pytest.fixture(scope="module")
def world():
    return make_vlq()


def test_graph_build_matches_oracle(world):
    v, _, _ = world
    g = gpu_from_oracle(v, with_lists=False, graph=False)
    ei, ed = g.build_graph()
    assert np.array_equal(ei, v.edge_info)
    assert np.array_equal(bits(ed), bits(v.edge_dist))


def test_assign_and_encode_match_oracle(world):
    v, xb, _ = world
    g = gpu_from_oracle(v, with_lists=False)
    line, lam = g.assign(xb)
    lo, lamo = v.assign(xb)
    assert np.array_equal(line, lo)
    assert np.array_equal(bits(lam), bits(lamo))
    lb = v.quantize_lambda(lamo)
    assert np.array_equal(bits(g.residuals(xb)), bits(v.residuals(xb, lo, lb)))
    l2, lb2, codes = g.encode(xb)
    lo2, lbo, co = v.encode(xb)
    assert np.array_equal(l2, lo2) and np.array_equal(lb2, lbo) and np.array_equal(codes, co)


def test_add_builds_the_same_lines(world):
    v, xb, _ = world
    g = gpu_from_oracle(v, with_lists=False)
    g.add(xb[:1500])
    g.add(xb[1500:])
    assert g.ntotal == v.ids.shape[0]
    for line in range(v.nlist * v.nedge):
        c, l, i = g.get_list(line)
        o0, o1 = v.line_off[line], v.line_off[line + 1]
        assert np.array_equal(i, v.ids[o0:o1]) and np.array_equal(c, v.codes[o0:o1]) and np.array_equal(l, v.lambdas[o0:o1])


@pytest.mark.parametrize("nprobe,w1,k", [(8, 24, 10), (4, 1, 1), (24, 144, 100), (6, 30, 300), (24, 64, 64)])
def test_search_bit_exact_vs_oracle(world, nprobe, w1, k):
    v, _, xq = world
    g = gpu_from_oracle(v)
    D, I, lines = g.search(xq, nprobe, w1, k, return_lines=True)
    Do, Io, lo = v.search(xq, nprobe, w1, k, return_lines=True)
    assert np.array_equal(lines, lo)
    assert np.array_equal(bits(D), bits(Do))
    assert np.array_equal(I, Io)
    assert g.stats(reset=True) == v.last_ncode


def test_search_distances_are_true_distances_minus_query_norm(world):
    v, _, xq = world
    g = gpu_from_oracle(v)
    D, I = g.search(xq, 8, 24, 5)
    pos_of = {int(i): p for p, i in enumerate(v.ids)}
    for qi in range(10):
        q = xq[qi].astype(np.float64)
        for j in range(5):
            y = decode(v, pos_of[int(I[qi, j])])
            ref = ((q - y) ** 2).sum() - (q ** 2).sum()
            assert abs(D[qi, j] - ref) <= 1e-3 * max(1.0, abs(ref))


def test_line_cap_1024_codes():
    """Lines are scanned up to 1024 codes (PQScanMultiPassPrecomputed.cu:728)."""
    v, xb, xq = make_vlq(seed=3, nlist=4, nedge=2, nb=6000)
    assert np.diff(v.line_off).max() > 1024
    g = gpu_from_oracle(v)
    D, I = g.search(xq, 4, 8, 20)
    Do, Io = v.search(xq, 4, 8, 20)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)
    assert g.stats() == v.last_ncode < xq.shape[0] * 6000


@pytest.mark.parametrize("rows", [1, 2, 3])
def test_m16_8bit_d128_shape(rows):
    """The 16-byte-code shape of the reference's deep1b16 / sift1b16 drivers; term-2 rows read from the
    stored table (line16_scan_kernel), rebuilt in registers (line16r_scan_kernel, dsub = 8), and the scan with
    the stored per-code constants (line16c_scan_kernel, rows = 3 = what 0 / automatic resolves to)."""
    v, xb, xq = make_vlq(seed=5, d=128, nlist=32, M=16, nbits=8, nedge=8, nlambda=64, nb=4000)
    g = gpu_from_oracle(v)
    g.set_row_mode(rows)
    D, I = g.search(xq, 8, 32, 10)
    Do, Io = v.search(xq, 8, 32, 10)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)


@pytest.fixture(scope="module")
def world16():
    """16-byte codes, enough lines for w1 = 1024, many empty and a few over-long lines."""
    return make_vlq(seed=11, d=96, nlist=200, M=16, nbits=8, nedge=8, nlambda=256, nb=20000)


@pytest.mark.parametrize("rows", [1, 2, 3])
@pytest.mark.parametrize("nprobe,w1,k", [(8, 32, 10), (16, 128, 128), (64, 300, 300), (8, 64, 1),
                                         (128, 1024, 128), (200, 1024, 1000), (3, 5, 64), (64, 1024, 256)])
def test_m16_scan_kernel_bit_exact(world16, nprobe, w1, k, rows):
    """line16_scan_kernel (anchor-grouped lines, compact line records, two-table gathers; rows = 1) and
    line16r_scan_kernel (term-2 rows rebuilt from the centroid and the codebook, interleaved double-buffered
    table; rows = 2, k <= 256 -- larger selections fall back to the stored rows) and line16c_scan_kernel (one table
    per anchor, the query-independent la * sum(term 4) read as a stored per-code constant; rows = 3) against the
    oracle over the selection sizes of the wave select (w1 -> 1/4/16 keys per lane, k likewise)."""
    v, _, xq = world16
    g = gpu_from_oracle(v)
    g.set_row_mode(rows)
    D, I, lines = g.search(xq, nprobe, w1, k, return_lines=True)
    Do, Io, lo = v.search(xq, nprobe, w1, k, return_lines=True)
    assert np.array_equal(lines, lo)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)
    assert g.stats() == v.last_ncode


@pytest.mark.parametrize("parts", [2, 3, 7, 64])
@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("nprobe,w1,k", [(8, 32, 10), (64, 300, 300), (128, 1024, 128), (3, 5, 64), (200, 1024, 1000)])
def test_m16_scan_in_parts_is_the_same_scan(world16, nprobe, w1, k, fp16, parts):
    """line16c_scan_kernel with a query's kept codes cut into ranges scanned by separate workgroups (cuts fall inside
    lines and inside anchor groups; with 64 parts some are empty) + line16c_merge_kernel: the same rows, bit for bit."""
    v, _, xq = world16
    g = gpu_from_oracle(v)
    g.set_float16_tables(fp16)
    g.set_scan_parts(parts)
    D, I = g.search(xq, nprobe, w1, k)
    Do, Io = v.search(xq, nprobe, w1, k, fp16=fp16)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)
    assert g.stats() == v.last_ncode


@pytest.mark.parametrize("rows", [1, 2, 3])
def test_m16_line_cap_and_big_batch(rows):
    v, xb, xq = make_vlq(seed=13, d=128, nlist=6, M=16, nbits=8, nedge=2, nlambda=32, nb=9000)
    assert np.diff(v.line_off).max() > 1024
    g = gpu_from_oracle(v)
    g.set_row_mode(rows)
    xq = np.concatenate([xq] * 30)          # 1200 queries: several waves of workgroups
    D, I = g.search(xq, 6, 12, 20)
    Do, Io = v.search(xq, 6, 12, 20)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)


def test_error_paths(world):
    v, _, xq = world
    g = vlq.GpuVLQ(v.d, v.nlist, v.M, v.nbits, v.nedge, v.nlambda)
    with pytest.raises(vlq.VlqError):
        g.search(xq, 4, 8, 5)            # nothing trained
    with pytest.raises(vlq.VlqError):
        vlq.GpuVLQ(v.d, v.nlist, v.M, v.nbits, v.nlist, 16)     # nedge >= nlist
    with pytest.raises(vlq.VlqError):
        vlq.GpuVLQ(v.d, v.nlist, v.M, v.nbits, 4, 300)          # lambda index does not fit a byte


import os


@pytest.mark.parametrize("seed", range(int(os.environ.get("VLQ_FUZZ_SEEDS_VLQ", "24"))))   # e.g. 300 for a soak run
def test_random_vlq_configuration(seed):
    """Seeded random VLQ shapes (generic and 16-byte scan kernels, device-side add in batches,
    empty and over-long lines, w1 / k on both sides of the selection widths) against the oracle."""
    rng = np.random.default_rng(500 + seed)
    M = int(rng.choice([2, 4, 8, 16, 16, 32]))     # 2: generic scan; 4 / 8 / 32 with small tables: lineS; 16 x 8: line16
    nbits = 8 if M == 16 else (int(rng.choice([4, 5])) if M == 32 else int(rng.choice([4, 6, 8])))
    dsub = int(rng.choice([2, 4, 6, 8]))
    nlist = int(rng.choice([6, 20, 64]))
    nedge = int(rng.choice([1, 2, 5]))
    nedge = min(nedge, nlist - 1)
    nlambda = int(rng.choice([4, 32, 256]))
    nb = int(rng.choice([50, 1500, 6000]))
    v, xb, xq = make_vlq(seed=600 + seed, d=M * dsub, nlist=nlist, M=M, nbits=nbits, nedge=nedge, nlambda=nlambda, nb=nb)
    g = gpu_from_oracle(v, with_lists=False)
    cut = nb // 3
    g.add(xb[:cut])
    if M == 16 and seed % 2 == 0:      # a search between the adds: the per-code constants must be rebuilt after the second
        g.search(xq[:4], 4, 7, 10)
    g.add(xb[cut:])
    assert g.ntotal == nb
    for line in range(nlist * nedge):
        c, lam, ids = g.get_list(line)
        o0, o1 = v.line_off[line], v.line_off[line + 1]
        assert np.array_equal(ids, v.ids[o0:o1]) and np.array_equal(c, v.codes[o0:o1]) and np.array_equal(lam, v.lambdas[o0:o1])
    nprobe = int(rng.choice([1, 4, nlist]))
    w1 = int(rng.choice([1, 7, 64, 65, 300]))
    k = int(rng.choice([1, 10, 64, 65, 257]))
    D, I, lines = g.search(xq, nprobe, w1, k, return_lines=True)
    Do, Io, lo = v.search(xq, nprobe, w1, k, return_lines=True)
    assert np.array_equal(lines, lo)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)
    if M == 16:        # the default above is the per-code-constant scan; rows read from the stored table (1) and rebuilt
        for rows in (1, 2):   # in the kernel (2: dsub 4 / 6 / 8, k <= 256; other shapes keep the stored rows) must agree
            g.set_row_mode(rows)
            D1, I1 = g.search(xq, nprobe, w1, k)
            assert np.array_equal(bits(D1), bits(Do)) and np.array_equal(I1, Io)


def test_constants_follow_incremental_adds_and_a_change_of_table_precision(world16):
    """The stored per-code constants of line16c: a small add that fits the lists' slack appends in place and
    extends the constants by the new vectors only; one that does not rebuilds the layout and with it the constants; toggling the
    tables' precision keeps one precision's buffer.  A search after every step equals the oracle's over the same prefix."""
    v, xb, xq = world16
    nb = xb.shape[0]
    g = gpu_from_oracle(v, with_lists=False)
    steps = [int(nb * 0.70), int(nb * 0.72), int(nb * 0.73), nb]          # the two middle adds fit the 25 % slack
    done = 0
    for i, upto in enumerate(steps):
        g.add(xb[done:upto])
        done = upto
        keep = v.ids < done                                              # (ids are 0 .. nb-1 in add order)
        sub = v.restricted_to(keep) if hasattr(v, "restricted_to") else None
        fp16 = i == 2                                                    # float16 tables for one step, fp32 again after
        g.set_float16_tables(fp16)
        D, I = g.search(xq[:64], 16, 128, 20)
        if sub is not None:
            Do, Io = sub.search(xq[:64], 16, 128, 20, fp16=fp16)
            assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io), i
        else:
            # no oracle over a prefix: the same prefix loaded in one piece into a fresh index must answer alike
            h = gpu_from_oracle(v, with_lists=False)
            h.add(xb[:done])
            h.set_float16_tables(fp16)
            D2, I2 = h.search(xq[:64], 16, 128, 20)
            assert np.array_equal(bits(D), bits(D2)) and np.array_equal(I, I2), i
    g.set_float16_tables(False)
    D, I = g.search(xq, 16, 128, 20)
    Do, Io = v.search(xq, 16, 128, 20)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)


# ---------------------------------------------------------------------------------------------
# float16 look-up tables (GpuIndexIVFPQConfig::useFloat16LookupTables, the reference drivers' setting)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows", [1, 3])
@pytest.mark.parametrize("nprobe,w1,k", [(8, 32, 10), (16, 128, 128), (64, 300, 300), (128, 1024, 128), (200, 1024, 1000)])
def test_fp16_tables_bit_exact_vs_fp16_oracle(world16, nprobe, w1, k, rows):
    """half(term 2), half(term 3), half add / subtract for the two tables, float accumulation: the
    device must reproduce the oracle's float16 mode bit for bit (same order, same ties)."""
    v, _, xq = world16
    g = gpu_from_oracle(v)
    g.set_row_mode(rows)              # 1: line16h_scan_kernel; 3: line16c_scan_kernel with the half-table constants
    g.set_float16_tables(True)
    D, I, lines = g.search(xq, nprobe, w1, k, return_lines=True)
    Do, Io, lo = v.search(xq, nprobe, w1, k, return_lines=True, fp16=True)
    assert np.array_equal(lines, lo)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)
    g.set_float16_tables(False)                       # and back: the fp32 tables are untouched
    D32, I32 = g.search(xq, nprobe, w1, k)
    Do32, Io32 = v.search(xq, nprobe, w1, k)
    assert np.array_equal(bits(D32), bits(Do32)) and np.array_equal(I32, Io32)


def test_fp16_tables_within_the_reference_gpu_vs_cpu_bar(world16):
    """gpu/test/TestGpuIndexIVFPQ.cpp:89-99: relative distance error <= 0.015 and <= 10 % differing
    results between float16 tables and the fp32 answer (distances compared on the |q - y|^2 scale)."""
    v, _, xq = world16
    g = gpu_from_oracle(v)
    D32, I32 = g.search(xq, 32, 256, 50)
    g.set_float16_tables(True)
    D16, I16 = g.search(xq, 32, 256, 50)
    qn = (xq.astype(np.float64) ** 2).sum(1)[:, None]
    ok = (I32 >= 0) & (I16 == I32)
    rel = np.abs(D16.astype(np.float64) - D32)[ok] / np.maximum(D32[ok] + np.broadcast_to(qn, D32.shape)[ok], 1e-9)
    assert rel.max() <= 0.015
    assert (I16 == I32).mean() >= 0.9


def test_fp16_tables_only_for_16_byte_codes(world):
    v, _, _ = world
    g = gpu_from_oracle(v)            # M = 8 x 6 bit
    with pytest.raises(vlq.VlqError):
        g.set_float16_tables(True)


def test_config4_literal_shape_m32_4bit():
    """BASELINE.json configs[4] as written: 32 sub-quantizers x 4 bits, nprobe 256, recall@100 (k = 100) --
    served by the generic line scan (one byte per sub-quantizer index, as the reference stores codes)."""
    v, xb, xq = make_vlq(seed=44, d=96, nlist=300, M=32, nbits=4, nedge=6, nlambda=64, nb=12000)
    g = gpu_from_oracle(v)
    D, I, lines = g.search(xq, 256, 1024, 100, return_lines=True)
    Do, Io, lo = v.search(xq, 256, 1024, 100, return_lines=True)
    assert np.array_equal(lines, lo)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)
    assert g.stats(reset=True) == v.last_ncode
