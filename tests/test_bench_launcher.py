"""CPU: bench.py's multi-rank plumbing with a stub index (BENCH_STUB=1, gloo) -- the self-launch of
`python bench.py --gpus N` without a launcher, the strong-scaling slices (IndexProxy.cpp:139-149), the
packed all-gather and the assertions that turn a wrong rank count or a wrong gather into a non-zero
exit.  The GPU suite runs the same code path with the HIP index (test_gpu_dist.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, **env):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(BENCH_STUB="1", OMP_NUM_THREADS="1", **env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, capture_output=True, text=True,
                       timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


@pytest.mark.parametrize("nq", [1000, 1001])          # even split and a ragged last slice
def test_self_launch_two_ranks_strong_and_weak(nq):
    rc, out, err = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--nq", str(nq), "--k", "7"])
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "strong"
    assert out["config"]["all_gather_check"] is True
    assert out["config"]["queries_per_rank"] == (nq + 1) // 2          # ceil(nq / N)
    assert out["other_scaling"]["scaling"] == "weak" and out["other_scaling"]["all_gather_check"] is True


def test_weak_scaling_can_be_the_headline():
    rc, out, err = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--nq", "300", "--scaling", "weak"])
    assert rc == 0, err[-2000:]
    assert out["scaling"] == "weak" and out["config"]["queries_per_rank"] == 300 and out["other_scaling"]["scaling"] == "strong"


def test_one_rank_needs_no_launcher_and_no_process_group():
    rc, out, err = run_bench(["--steps", "2", "--warmup", "1", "--nq", "100"])
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 1 and out["rccl_ranks"] is None


def test_rank_count_mismatch_is_an_error():
    rc, out, err = run_bench(["--gpus", "4", "--steps", "1", "--warmup", "0", "--nq", "50"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert rc != 0 and out is None and "WORLD_SIZE=1" in err


def test_wrong_gather_is_an_error():
    rc, out, err = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--nq", "400"], BENCH_STUB_CORRUPT="1")
    assert rc != 0 and out is None
    assert "all-gathered results differ" in err
