"""The reference-run helper of bench.py's CPU-baseline leg (oracle/refbench.py): an index written
in the reference's file format from fixture data, searched by the compiled reference itself,
must return the fixture's (= the reference's) own results.  Needs oracle/_ref (built by
oracle/ref.mk where /root/reference exists; travels to the GPU box as a built artefact)."""
import os

import numpy as np
import pytest

from oracle import refbench
from util import Case, bits

pytestmark = pytest.mark.skipif(not refbench.available(), reason="oracle/_ref not built")


@pytest.mark.parametrize("name", ["c1_small", "deep_like_dsub6"])
def test_reference_run_reproduces_fixture(tmp_path, name):
    case = Case(name)
    path = str(tmp_path / "ix.faissindex")
    refbench.write_ivfpq_index(path, case["coarse_centroids"], case["pq_centroids"], case.nbits, case["codes"],
                               case["ids"], case["list_offsets"])
    D, I, secs, meta = refbench.run_reference(path, case.xq, case.nprobe, case.k, reps=1, threads=2)
    assert meta[0] == case.mode and meta[1] == int(case["ncode"][0])
    assert np.array_equal(bits(D), bits(case["D"]))
    assert np.array_equal(I, case["I"])
    assert secs.shape == (1,) and secs[0] > 0


def test_written_file_is_what_the_reference_writes(tmp_path):
    case = Case("tiny_padding")           # this fixture carries the reference's own write_index bytes
    path = str(tmp_path / "ix.faissindex")
    refbench.write_ivfpq_index(path, case["coarse_centroids"], case["pq_centroids"], case.nbits, case["codes"],
                               case["ids"], case["list_offsets"], nprobe=case.nprobe)
    assert np.array_equal(np.fromfile(path, dtype=np.uint8), case["faissindex_file"])
