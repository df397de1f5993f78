"""Query sharding across the GPUs of one node.

The search path has no cross-query state (per-query beams, per-query evidence;
reference seal/retrieval.py:795-804), so it shards embarrassingly: one process
per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo"
in CPU tests), FM-index and model replicated on every GPU, each rank searches a
contiguous block of the query batch.  The ONLY exchange is the final top-k
gather: fixed shape ``[queries_per_rank, k, 2]`` float64 (doc id, score) per rank,
~1.6 KB per query at k=100 -- latency-bound, one ``all_gather``, no ring tuning.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous block of ``n_items`` owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_queries(queries: Sequence, rank: int = None, world: int = None) -> List:
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_bounds(len(queries), rank, world)
    return list(queries[lo:hi])


def pack_topk(results, k: int) -> torch.Tensor:
    """``[[SEALDocument...]...]`` (or ``[[(doc, score)...]...]``) -> float64 ``[n, k, 2]``, padded with -1."""
    import numpy as np
    out = np.full((len(results), k, 2), -1.0, dtype=np.float64)
    for qi, docs in enumerate(results):
        docs = docs[:k]
        if docs:
            out[qi, :len(docs)] = [(d.idx, d.score) if hasattr(d, "idx") else d for d in docs]
    return torch.from_numpy(out)


def gather_topk(local: torch.Tensor, n_total: int, device=None) -> torch.Tensor:
    """all-gather the per-rank ``[n_local, k, 2]`` blocks into ``[n_total, k, 2]`` in query order.
    Ranks may own different numbers of queries (``shard_bounds``); blocks are padded to the
    largest shard for the collective and trimmed afterwards."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    k = local.shape[1]
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = torch.full((pad, k, 2), -1.0, dtype=torch.float64, device=device or local.device)
    buf[:local.shape[0]] = local.to(buf.device)
    out = torch.empty(world * pad, k, 2, dtype=torch.float64, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    out = out.view(world, pad, k, 2)
    return torch.cat([out[r, :hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)


def sharded_batch_search(searcher, queries: Sequence, k: int = 100, device=None) -> torch.Tensor:
    """``SEALSearcher.batch_search`` over the ranks of the default process group: this rank searches its contiguous
    block of ``queries`` (``shard_bounds``), then ONE all-gather brings every query's top-k (doc id, score) to every rank,
    in query order: float64 ``[len(queries), k, 2]``, -1 where a query has fewer than k hits.  Without a process group it
    is a plain local search."""
    mine = shard_queries(queries)
    results = searcher.batch_search(mine, k=k, detokenize=False) if mine else []
    local = pack_topk(results, k)
    if device is not None:
        local = local.to(device)
    return gather_topk(local, len(queries), device=device)
