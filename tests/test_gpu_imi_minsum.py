"""GPU: the MinSumK replay of the multi-index coarse quantizer (IndexPQ.cpp:690-778) -- the wave-per-query walk with the heap
in registers (kernels.hip::imi_minsum_wave_kernel, nprobe <= 64) against the oracle, bit for bit.  The walk's values are path
dependent and its order under EQUAL sums is the binary heap's (Heap.h:89-127), so the data here sit on an integer grid: sums
tie in long runs, cells are reached from both neighbours with equal values, and the heap's positions decide.  Small tables
(fewer than nprobe cells), every nprobe up to 64, and the thread-per-query kernel (VLQ_IMI_MINSUM_LDS) on the same inputs in a
second process."""
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

bad, ncase = [], 0
for nbits, dc, grid in ((1, 4, 3), (2, 4, 3), (3, 8, 2), (4, 4, 4), (6, 8, 3), (8, 16, 2), (10, 8, 1000)):
    rng = np.random.default_rng(1000 * nbits + dc)
    kc, d, M = 1 << nbits, 2 * dc, 2
    # grid = 1000: ordinary data; small grids: coordinates in {0 .. grid-1} -> integer distances, many equal
    draw = (lambda shape: rng.integers(0, grid, shape).astype(np.float32)) if grid < 1000 else (lambda shape: rng.random(shape, dtype=np.float32))
    imi = draw((2, kc, dc))
    pq = rng.random((M, 256, d // M), dtype=np.float32)
    xq = draw((300, d))
    m = min(8, kc)
    xq[:m] = np.concatenate([imi[0, :m], imi[1, :m]], axis=1)          # queries ON cells: zeros and ties at the front
    g = vlq.GpuIVFPQ(d, kc * kc, M, 8)
    g.set_imi_centroids(nbits, imi)
    g.set_pq_centroids(pq)
    ox = OracleIndex(d, kc * kc, M, 8, None, pq, imi_centroids=imi, imi_nbits=nbits)
    for nprobe in (2, 3, 4, 5, 7, 8, 16, 17, 31, 32, 33, 48, 63, 64):
        if nprobe > kc * kc:
            continue
        cd, keys = g.coarse_search(xq, nprobe)
        cdo, keyso = ox.coarse_search(xq, nprobe, canonical=True)
        ncase += 1
        if not (np.array_equal(keys, keyso) and np.array_equal(bits(cd), bits(cdo))):
            bad.append((nbits, dc, grid, nprobe, int((keys != keyso).any(axis=1).sum())))
print("CASES", ncase, "BAD", bad)
assert not bad and ncase > 60, (ncase, bad)
"""


@pytest.mark.parametrize("name,extra", [("wave_per_query", {}), ("thread_per_query", {"VLQ_IMI_MINSUM_LDS": "1"})])
def test_minsum_walk_equals_the_oracle_on_tied_sums(name, extra):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(extra)
    p = subprocess.run([sys.executable, "-c", CODE], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (name, p.stdout[-2000:], p.stderr[-3000:])
