"""CPU, world_size 2, gloo: the N>1 path (query sharding + all-gather of per-shard
top-k).  The local search on CPU ranks is the oracle (test infrastructure); on a GPU
box the same function is driven with the HIP index by bench.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vector_line_quantization_amd.sharded import (list_range, list_sharded_search, merge_shard_results,
                                                  shard_bounds, sharded_search)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 10000, 10001):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi, per = shard_bounds(n, world, r)
                assert 0 <= lo <= hi <= n and hi - lo <= per
                covered += list(range(lo, hi))
            assert covered == list(range(n))


def _worker(rank, world, port, nq_used, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from util import Case
    case = Case("c1_small")
    ox = case.oracle_index()

    def local_search(xs, nprobe, k):
        D, I = ox.search(xs.numpy(), nprobe, k, canonical=True)
        return torch.from_numpy(D), torch.from_numpy(I)

    x = torch.from_numpy(case.xq[:nq_used])
    D, I = sharded_search(local_search, x, case.nprobe, case.k)
    np.save(os.path.join(out_dir, "D%d.npy" % rank), D.numpy())
    np.save(os.path.join(out_dir, "I%d.npy" % rank), I.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq_used", [64, 51])   # even split and ragged last shard
def test_query_sharded_search_world2(tmp_path, nq_used):
    from util import Case
    world = 2
    port = 29500 + (os.getpid() + nq_used) % 2000
    mp.spawn(_worker, args=(world, port, nq_used, str(tmp_path)), nprocs=world, join=True)
    case = Case("c1_small")
    ox = case.oracle_index()
    Dref, Iref = ox.search(case.xq[:nq_used], case.nprobe, case.k, canonical=True)
    for r in range(world):   # every rank holds the full, identical result
        D = np.load(tmp_path / ("D%d.npy" % r))
        I = np.load(tmp_path / ("I%d.npy" % r))
        assert np.array_equal(D.view(np.uint32), Dref.view(np.uint32))
        assert np.array_equal(I, Iref)


def test_merge_shard_results_matches_global_topk():
    rng = np.random.default_rng(0)
    n, k, parts = 17, 5, 3
    D = np.sort(rng.random((parts, n, k)).astype(np.float32), axis=2)
    I = rng.integers(0, 10 ** 6, (parts, n, k))
    Dm, Im = merge_shard_results([torch.from_numpy(d) for d in D], [torch.from_numpy(i) for i in I], k)
    allD = np.concatenate(list(D), axis=1)
    allI = np.concatenate(list(I), axis=1)
    order = np.argsort(allD, axis=1, kind="stable")[:, :k]
    assert np.array_equal(Dm.numpy(), np.take_along_axis(allD, order, 1))
    assert np.array_equal(Im.numpy(), np.take_along_axis(allI, order, 1))


def _worker_lists(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from util import Case
    case = Case("c1_small")
    ox = case.oracle_index()
    # this rank keeps only the lists of its range: all other lists become empty
    lo, hi = list_range(case.nlist, world, rank)
    off = ox.list_offsets
    keep = np.zeros(ox.ids.shape[0], bool)
    keep[off[lo]:off[hi]] = True
    lens = np.diff(off)
    lens[:lo] = 0
    lens[hi:] = 0
    new_off = np.zeros_like(off)
    np.cumsum(lens, out=new_off[1:])
    ox.set_lists(ox.codes[keep], ox.ids[keep], new_off)

    def local_search(xs, nprobe, k):
        D, I = ox.search(xs.numpy(), nprobe, k, canonical=True)
        return torch.from_numpy(D), torch.from_numpy(I)

    D, I = list_sharded_search(local_search, torch.from_numpy(case.xq), case.nprobe, case.k)
    np.save(os.path.join(out_dir, "LD%d.npy" % rank), D.numpy())
    np.save(os.path.join(out_dir, "LI%d.npy" % rank), I.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_list_sharded_search_world2(tmp_path):
    """List-range sharding + merge returns the distances of the unsharded index (labels
    may permute inside exact-distance ties: the merge orders ties by rank)."""
    from util import Case, assert_same_topk
    world = 2
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_lists, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    case = Case("c1_small")
    Dref, Iref = case.oracle_index().search(case.xq, case.nprobe, case.k, canonical=True)
    for r in range(world):
        D = np.load(tmp_path / ("LD%d.npy" % r))
        I = np.load(tmp_path / ("LI%d.npy" % r))
        assert_same_topk(D, I, Dref, Iref, "list-sharded")
