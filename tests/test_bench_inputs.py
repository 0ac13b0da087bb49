"""CPU: bench.py's .fvecs / .ivecs reader (the reference driver's format, tests/demo_sift1M.cpp:40-70) and
the discovery of a SIFT1M directory."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def write_vecs(path, a):
    n, d = a.shape
    out = np.empty((n, d + 1), np.int32)
    out[:, 0] = d
    out[:, 1:] = a.view(np.int32)
    out.tofile(path)


def test_fvecs_and_ivecs_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    x = rng.random((37, 12), dtype=np.float32)
    gt = rng.integers(0, 1000, (37, 5)).astype(np.int32)
    write_vecs(tmp_path / "x.fvecs", x)
    write_vecs(tmp_path / "gt.ivecs", gt)
    assert np.array_equal(bench.fvecs_read(str(tmp_path / "x.fvecs")), x)
    assert np.array_equal(bench.fvecs_read(str(tmp_path / "gt.ivecs")).view(np.int32), gt)


def test_sift_directory_is_found_only_when_complete(tmp_path):
    assert bench.find_fvecs_dir(str(tmp_path)) in (None, os.environ.get("SIFT1M_DIR"), "/home/data/sift1m", os.path.join(ROOT, "data", "sift1m"))
    x = np.zeros((2, 4), np.float32)
    for f in ("learn.fvecs", "base.fvecs"):
        write_vecs(tmp_path / f, x)
    assert bench.find_fvecs_dir(str(tmp_path)) != str(tmp_path)
    write_vecs(tmp_path / "query.fvecs", x)
    assert bench.find_fvecs_dir(str(tmp_path)) == str(tmp_path)
