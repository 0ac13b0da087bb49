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


def test_list_sharded_search_with_vlq_handles_holding_line_ranges(tmp_path):
    """The fork's MPI mode (gpu/GpuIndexIVFPQ.cu:2106-2242, gpu/test/deep1b16_query.cpp:193-428) through the Python
    host: two ranks (two processes sharing the box's one GPU, gloo for the exchange), each a GpuVLQ that holds
    the LINES of its rank's range only -- all other lines empty -- searches every query; list_sharded_search
    all-gathers and merges the per-rank rows.  Distances are those of the unsharded VLQ search bit for bit,
    labels up to exact-distance ties (the merge orders ties by rank)."""
    code = r"""
import os, sys
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch, torch.distributed as dist
import vector_line_quantization_amd as vlq
from vector_line_quantization_amd.sharded import list_range, list_sharded_search
from test_vlq_oracle import make_vlq
dist.init_process_group('gloo', rank=rank, world_size=world)
v, xb, xq = make_vlq(seed=21, d=96, nlist=64, M=16, nbits=8, nedge=4, nlambda=64, nb=8000)
nl = v.nlist * v.nedge
lo, hi = list_range(nl, world, rank)          # readDbFromFile(name, pronum, rank): lines [lo, hi) of this rank
off = v.line_off
lens = np.diff(off); lens[:lo] = 0; lens[hi:] = 0
new_off = np.zeros_like(off); np.cumsum(lens, out=new_off[1:])
keep = slice(off[lo], off[hi])
g = vlq.GpuVLQ(v.d, v.nlist, v.M, v.nbits, v.nedge, v.nlambda)
g.set_coarse_centroids(v.coarse); g.set_graph(v.edge_info, v.edge_dist)
g.set_lambda_codebook(v.lambda_info); g.set_pq_centroids(v.pq_centroids)
g.set_lists(v.codes[keep], v.lambdas[keep], v.ids[keep], new_off)
assert g.ntotal == int(lens.sum()) < v.ids.shape[0]
def local_search(xs, nprobe, k):
    D, I = g.search(xs.numpy(), nprobe, 48, k)
    return torch.from_numpy(D), torch.from_numpy(I)
D, I = list_sharded_search(local_search, torch.from_numpy(xq), 16, 20)
np.save(os.path.join(out, 'D%%d.npy' %% rank), D.numpy()); np.save(os.path.join(out, 'I%%d.npy' %% rank), I.numpy())
if rank == 0:
    Do, Io = v.search(xq, 16, 48, 20)
    np.save(os.path.join(out, 'Do.npy'), Do); np.save(os.path.join(out, 'Io.npy'), Io)
dist.barrier(); dist.destroy_process_group()
print('ok')
""" % (ROOT, ROOT)
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import assert_same_topk
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29620 + os.getpid() % 300), OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2", str(tmp_path)], env=e, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "ok" in so, se[-3000:]
    Do, Io = np.load(tmp_path / "Do.npy"), np.load(tmp_path / "Io.npy")
    for r in range(2):
        assert_same_topk(np.load(tmp_path / ("D%d.npy" % r)), np.load(tmp_path / ("I%d.npy" % r)), Do, Io, "vlq line-sharded")
