"""GPU: the HIP index inside a process group.  bench.py with BENCH_FORCE_DIST=1 takes the N > 1 code
path (RCCL process group, packed all-gather on the collective stream overlapping the next search,
asserted gather check) with one rank -- all a one-GPU box can run; tests/test_bench_launcher.py covers
the N = 2 slicing and launcher logic on CPU, tests/test_sharded_gloo.py the library function."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_collective_path_with_hip_index():
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--nb", "200000", "--nt", "50000", "--nlist", "1024", "--no-cpu-baseline", "--no-second-dataset",
                        "--no-host-buffers", "--no-vlq"],
                       env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["rccl_ranks"] == 1 and out["n_gpus"] == 1
    assert out["config"]["all_gather_check"] is True
    assert out["parity"]["distance_bits_equal"] is True and out["parity"]["label_mismatches"] == 0
    assert out["roofline"]["kernel_ms"] > 0 and "post_run_error" not in out


def test_sharded_search_function_with_hip_index():
    """vector_line_quantization_amd.sharded.sharded_search driven by the HIP index (one rank, RCCL
    backend): the function the gloo tests drive with the oracle."""
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch, torch.distributed as dist
import vector_line_quantization_amd as vlq
from vector_line_quantization_amd.sharded import sharded_search, list_sharded_search
from util import Case
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
case = Case('c1_small')
g = vlq.GpuIVFPQ(case.d, case.nlist, case.M, case.nbits)
g.set_stream(torch.cuda.current_stream().cuda_stream)
g.set_coarse_centroids(case['coarse_centroids']); g.set_pq_centroids(case['pq_centroids'])
g.set_lists(case['codes'], case['ids'], case['list_offsets'])
x = torch.from_numpy(case.xq).cuda()
for fn in (sharded_search, list_sharded_search):
    D, I = fn(lambda xs, nprobe, k: g.search(xs, nprobe, k), x, case.nprobe, case.k)
    torch.cuda.synchronize()
    Do, Io = case.oracle_index().search(case.xq, case.nprobe, case.k, canonical=True)
    assert np.array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32)) and np.array_equal(I.cpu().numpy(), Io)
dist.destroy_process_group()
print('ok')
""" % (ROOT, ROOT)
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29612", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-3000:]
