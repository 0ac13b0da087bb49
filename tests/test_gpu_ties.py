"""GPU: equal distances.  The reference's heap admits `dis < top` while it scans a query's lists in coarse-distance order
(IndexIVFPQ.cpp:983-1060, Heap.h:76-78), so among codes at the same distance the one scanned FIRST stays.  The scan kernels
visit a query's lists, and the chunks of one list, in another order (walk_order.cuh; the chunks requested a probe ahead come
after a long list's further chunks) and select on (distance, scan position) keys: a code that ties with the current k-th
distance but sits EARLIER in the reference's scan has to replace it.  Lists full of identical codes, identical lists under
identical centroids, every code size, the three batch classes (split over workgroups, four waves, two waves), against the
oracle bit for bit -- ids included."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CODE = r"""
import sys, numpy as np
sys.path.insert(0, "tests")
import vector_line_quantization_amd as vlq
from oracle.pyoracle import OracleIndex
from util import bits

def make(M, nlist, seed, twins):
    rng = np.random.default_rng(seed)
    dsub = 4
    d = M * dsub
    coarse = rng.random((nlist, d)).astype(np.float32)
    if twins:                                   # lists under IDENTICAL centroids: equal distances across lists
        coarse[1::2] = coarse[0::2]
    pq = (0.2 * rng.standard_normal((M, 256, dsub))).astype(np.float32)
    lens = rng.integers(150, 700, nlist)
    lens[3 % nlist] = 1500                      # longer than everything a workgroup requests a probe ahead
    lens[6 % nlist] = 2300
    off = np.zeros(nlist + 1, np.int64)
    off[1:] = np.cumsum(lens)
    codes = rng.integers(0, 256, (off[-1], M), dtype=np.uint8)
    few = rng.integers(0, 256, (3, M), dtype=np.uint8)
    for l in range(nlist):
        s, e = off[l], off[l + 1]
        if l % 4 == 3:
            codes[s:e] = few[0]                 # one code, repeated
        elif l % 4 == 2:
            codes[s:e] = few[rng.integers(0, 3, e - s)]   # three codes
        elif twins and l % 2 == 1 and lens[l] <= lens[l - 1]:
            codes[s:e] = codes[off[l - 1]:off[l - 1] + lens[l]]   # the twin list repeats its neighbour's codes
    ids = rng.permutation(off[-1]).astype(np.int64)
    ox = OracleIndex(d, nlist, M, 8, coarse, pq)
    ox.set_lists(codes, ids, off)
    g = vlq.GpuIVFPQ(d, nlist, M, 8)
    g.set_coarse_centroids(coarse)
    g.set_pq_centroids(pq)
    g.set_lists(codes, ids, off)
    return rng, ox, g, d

bad = []
for M in (8, 12, 16, 24, 28, 32, 48):
    for twins in (False, True):
        for nlist, nprobe in ((32, 8), (2048, 16)):
            rng, ox, g, d = make(M, nlist, 17 * M + nlist + twins, twins)
            for nq in (40, 1500, 3100):
                xq = rng.random((nq, d)).astype(np.float32)
                for k in (1, 10, 100):
                    D, I = g.search(xq, nprobe, k)
                    Do, Io = ox.search(xq, nprobe, k, canonical=True)
                    if not (np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io)):
                        bad.append((M, twins, nlist, nq, k, int((I != Io).any(axis=1).sum())))
print("BAD", bad)
assert not bad, bad
"""


@pytest.mark.parametrize("name,extra", [("library", {}), ("all_by_id", {"VLQ_WALK_FIRST": "0", "VLQ_WALK_SHARE": "1000"}),
                                        ("one_first", {"VLQ_WALK_FIRST": "1", "VLQ_WALK_SHARE": "1000"}),
                                        ("reference_order", {"VLQ_WALK_FIRST": "-1"})])
def test_equal_distances_keep_the_reference_scan_order(name, extra):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(extra)
    p = subprocess.run([sys.executable, "-c", CODE], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (name, p.stdout[-2000:], p.stderr[-3000:])
