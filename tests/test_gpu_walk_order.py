"""GPU: the walking order of a query's probes (csrc/walk_order.cuh) and the workgroup shape of the 16-byte scan are schedules, not arithmetic: whatever order a workgroup
visits its lists in -- the reference's coarse-distance order (IndexIVFPQ.cpp:983-1060), the nearest n first and the rest by list
id, or the library's own per-batch decision -- distances, labels and the tie order must equal the oracle's, bit for bit.  The
library reads its A/B switches once per process, so every setting runs in a fresh process that checks itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CODE = r'''
import sys, json, numpy as np
sys.path.insert(0, "tests")
import test_gpu_code_sizes as t
from util import bits
out = {}
for M, dsub in ((16, 4), (32, 2), (8, 4)):
    rng, ox, g, gen = t.make(M, dsub, 160, 30000, 31 * M + dsub, long_frac=0.2)
    for nq, nprobe, k in ((1500, 32, 10), (1100, 16, 64), (1300, 150, 10), (1200, 40, 100), (300, 32, 10), (3100, 24, 20)):
        xq = gen(nq)
        xq[:10] = ox.coarse_centroids[:10]
        Do, Io = ox.search(xq, nprobe, k, canonical=True)
        for rep in range(3):       # from the second launch of a shape on, the walk starts where the workgroups' clock points
            D, I = g.search(xq, nprobe, k)
            assert np.array_equal(bits(D), bits(Do)), (M, nq, nprobe, k, rep)
            assert np.array_equal(I, Io), (M, nq, nprobe, k, rep)
        out["%d_%d_%d_%d" % (M, nq, nprobe, k)] = [int(bits(D).astype(np.uint64).sum()), int(I.sum())]
    # holes, invalid keys and the max_codes cut through the seam
    xq = gen(1400)
    cd, keys = g.coarse_search(xq, 24)
    keys = keys.copy()
    keys[rng.random(keys.shape) < 0.2] = -1
    keys[7] = -1
    g.set_search_options(max_codes=900)
    ox.max_codes = 900
    D, I = g.search_preassigned(xq, keys, cd, 10)
    Do, Io = ox.search_preassigned(xq, keys, cd, 10, canonical=True)
    assert np.array_equal(bits(D), bits(Do)) and np.array_equal(I, Io), ("seam", M)
print(json.dumps(out))
'''


@pytest.fixture(scope="module")
def runs():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, extra in (("library", {}), ("reference_order", {"VLQ_WALK_FIRST": "-1"}), ("all_by_id", {"VLQ_WALK_FIRST": "0"}),
                        ("one_first", {"VLQ_WALK_FIRST": "1"}), ("five_first", {"VLQ_WALK_FIRST": "5"}),
                        ("always_decide_id", {"VLQ_WALK_SHARE": "1000"}), ("fixed_clock", {"VLQ_WALK_SHARE": "1000", "VLQ_WALK_CLOCK": "777"}),
                        ("no_clock", {"VLQ_WALK_SHARE": "1000", "VLQ_WALK_CLOCK": "-1"}),
                        # the two workgroup shapes of the 16-byte kernel (two waves from 3000 queries on by default)
                        ("two_waves", {"VLQ_SCAN16_VARIANT": "4", "VLQ_WALK_SHARE": "1000"}), ("four_waves", {"VLQ_SCAN16_VARIANT": "1"})):
        env = dict(os.environ)
        env.update(extra)
        p = subprocess.run([sys.executable, "-c", CODE], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (name, p.stderr[-3000:])
        res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    return res


def test_every_walking_order_equals_the_oracle(runs):
    assert len(runs) == 10       # each process asserted its rows against the oracle


def test_walking_orders_agree_with_each_other(runs):
    ref = runs["reference_order"]
    for name, r in runs.items():
        assert r == ref, name
