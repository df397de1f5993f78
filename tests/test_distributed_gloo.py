"""world_size-2 gloo test of the N>1 path: query sharding + the final top-k gather
(the path's only collective)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from seal_amd.distributed import gather_topk, pack_topk, shard_bounds, shard_queries


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_queries, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    queries = list(range(n_queries))
    mine = shard_queries(queries)
    # a deterministic fake "search": query i returns docs i*100+j with scores 1000-i-j/10, fewer hits for odd i
    results = [[(i * 100 + j, 1000.0 - i - j / 10.0) for j in range(k if i % 2 == 0 else k // 2)] for i in mine]
    full = gather_topk(pack_topk(results, k), n_queries)
    q.put((rank, mine, full.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world, n_queries, k = 2, 7, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_queries, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort(key=lambda x: x[0])
    assert got[0][1] == [0, 1, 2, 3] and got[1][1] == [4, 5, 6]
    for _, _, full in got:            # every rank ends up with the whole, query-ordered result
        assert full.shape == (n_queries, k, 2)
        for i in range(n_queries):
            hits = k if i % 2 == 0 else k // 2
            for j in range(k):
                if j < hits:
                    assert full[i, j, 0].item() == i * 100 + j and abs(full[i, j, 1].item() - (1000.0 - i - j / 10.0)) < 1e-12
                else:
                    assert full[i, j, 0].item() == -1.0
    assert torch.equal(got[0][2], got[1][2])


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 20, 160):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
