"""Query-sharded multi-GPU search: index replicated on every rank, each rank
searches a contiguous slice of the query batch, one all-gather per output array
brings the per-shard top-k to every rank.

Re-expresses the reference's replica mode -- faiss::gpu::IndexProxy::search
(gpu/IndexProxy.cpp:123-168: slices of ceil(n/nreplica) queries, each replica
writes its slice of the caller's distances/labels) -- for one process per GPU with
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU
tests).  There is no cross-rank arithmetic, so results are bit-identical to a
single-GPU search by construction.  torch.distributed is plumbing here: the
search itself is whatever `local_search` is (the HIP index on a GPU box).
"""
import math


def shard_bounds(n, world, rank):
    """Slice [lo, hi) of rank `rank`: IndexProxy.cpp:139-149 arithmetic
    (queriesPerIndex = ceil(n / numIndices), last slices may be short or empty)."""
    per = int(math.ceil(n / float(world))) if n > 0 else 0
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi, per


def sharded_search(local_search, x, nprobe, k, group=None, D_out=None, I_out=None):
    """x: the FULL query batch [n, d] (torch tensor, same on every rank).
    local_search(xs, nprobe, k) -> (D[ns,k] float32, I[ns,k] int64) torch tensors on
    x's device.  Returns the full (D[n,k], I[n,k]) on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = x.shape[0]
    lo, hi, per = shard_bounds(n, world, rank)
    if world == 1:
        return local_search(x, nprobe, k)
    # equal-sized slots so that one all_gather_into_tensor moves everything
    Dl = torch.full((per, k), torch.finfo(torch.float32).max, dtype=torch.float32, device=x.device)
    Il = torch.full((per, k), -1, dtype=torch.int64, device=x.device)
    if hi > lo:
        d, i = local_search(x[lo:hi], nprobe, k)
        Dl[:hi - lo].copy_(d)
        Il[:hi - lo].copy_(i)
    Dall = torch.empty((world * per, k), dtype=torch.float32, device=x.device)
    Iall = torch.empty((world * per, k), dtype=torch.int64, device=x.device)
    dist.all_gather_into_tensor(Dall, Dl, group=group)
    dist.all_gather_into_tensor(Iall, Il, group=group)
    D = Dall[:n] if D_out is None else D_out.copy_(Dall[:n])
    I = Iall[:n] if I_out is None else I_out.copy_(Iall[:n])
    return D, I


def list_sharded_search(local_search, x, nprobe, k, group=None):
    """The reference's MPI mode (gpu/GpuIndexIVFPQ.cu:2106-2242, gpu/test/deep1b16_query.cpp:
    193-428) for indexes too large to replicate: every rank holds the inverted lists of a
    contiguous list range (all other lists empty), runs the FULL coarse stage for ALL queries
    (identical on every rank), scans only the probed lists it owns, then the per-rank top-k
    are all-gathered and merged (a select, never a sum).  x: full batch, same on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    D, I = local_search(x, nprobe, k)
    if world == 1:
        return D, I
    n = x.shape[0]
    Dall = torch.empty((world * n, k), dtype=torch.float32, device=x.device)
    Iall = torch.empty((world * n, k), dtype=torch.int64, device=x.device)
    dist.all_gather_into_tensor(Dall, D.contiguous(), group=group)
    dist.all_gather_into_tensor(Iall, I.contiguous(), group=group)
    Dall, Iall = Dall.view(world, n, k), Iall.view(world, n, k)
    if x.is_cuda:    # HIP merge kernel on the current stream
        import ctypes as C
        from ._lib import check, lib
        Dm = torch.empty((n, k), dtype=torch.float32, device=x.device)
        Im = torch.empty((n, k), dtype=torch.int64, device=x.device)
        check(lib().vlq_merge_topk(C.c_int(x.device.index or 0),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_int64(n),
                                   C.c_int(k), C.c_int(world), C.c_void_p(Dall.data_ptr()),
                                   C.c_void_p(Iall.data_ptr()), C.c_void_p(Dm.data_ptr()),
                                   C.c_void_p(Im.data_ptr())))
        return Dm, Im
    return merge_shard_results(list(Dall), list(Iall), k)


def list_range(nlist, world, rank):
    """Lists [r*L/P, (r+1)*L/P) of rank r (gpu/GpuIndexIVFPQ.cu:2132-2138)."""
    return rank * nlist // world, (rank + 1) * nlist // world


def merge_shard_results(D_parts, I_parts, k):
    """List-sharded mode (every rank holds a subset of the inverted lists and sees
    every query): merge per-rank top-k rows into the global top-k -- the role of
    GpuIndexIVFPQ::merge / mergekernel (gpu/GpuIndexIVFPQ.cu:1467-1591) and
    IndexShards' merge_tables (MetaIndexes.cpp:486-557).  Ties keep rank order."""
    import torch
    D = torch.cat(D_parts, dim=1)
    I = torch.cat(I_parts, dim=1)
    # padding (-1 / FLT_MAX) of shards with fewer than k hits sorts last by its distance
    order = torch.sort(D, dim=1, stable=True).indices[:, :k]
    return torch.gather(D, 1, order), torch.gather(I, 1, order)
