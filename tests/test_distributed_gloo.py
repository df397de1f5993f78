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
    q.put((rank, mine, full.numpy().tobytes(), tuple(full.shape)))     # by value: a tensor on a spawn queue travels as a file descriptor
    dist.barrier()                                                     # of a process that may be gone when the parent reads it
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
    import numpy as np
    got = sorted((r, mine, torch.from_numpy(np.frombuffer(raw, dtype=np.float64).reshape(shape).copy())) for r, mine, raw, shape in got)
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


# ---------------------------------------------------------------------------
# the real sharded search (not fake results): every rank runs the product's SEALSearcher.batch_search -- key generation,
# filters, rescoring, evidence aggregation -- on its block of the queries, then the top-k gather.  CPU: the index
# queries are answered by the oracle, as in tests/test_reference_golden.py.
# ---------------------------------------------------------------------------
def _cpu_searcher(batch_size, patch=None):
    import json
    from oracle.seal_oracle import OracleFMIndex
    from seal_amd import retrieval
    from seal_amd.retrieval import SEALSearcher
    from tests.helpers import OracleBatchIndex, OracleLogitsProcessor, tiny_bart
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_searcher.json")) as f:
        S = json.load(f)
    vocab, K, length, title_eos = S["vocab"], S["beam"], S["length"], S["title_eos"]
    orc = OracleFMIndex()
    orc.initialize(S["docs"])

    class CpuIndex(OracleBatchIndex):
        labels = None
        n_docs = property(lambda self: self.orc.n_docs)

        def get_doc(self, i):
            return self.orc.get_doc(i)
    real = retrieval.fm_index_generate

    def generate(model, _index, *a, **kw):
        proc = OracleLogitsProcessor(orc, kw["num_beams"], vocab, pad_token_id=1, eos_token_id=kw.get("eos_token_id") or 2,
                                     force_decoding_from=kw.get("force_decoding_from"))
        if kw.get("force_decoding_from"):
            kw = {**kw, "max_length": 8}
        return real(model, None, *a, constrained_decoding_processor=proc, **kw)
    if patch is not None:
        patch.setattr(retrieval, "fm_index_generate", generate)     # restored after the test (the main pytest process)
    else:
        retrieval.fm_index_generate = generate                       # a worker process of its own
    s = SEALSearcher(CpuIndex(orc), None, tiny_bart(vocab), backbone="bart-tiny", length=length, beam=K, batch_size=batch_size,
                     add_query_to_keys=True, detokenize=False, title_eos_token_id=title_eos, code_eos_token_id=vocab - 6,
                     code_bos_token_id=title_eos,
                     marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5]})
    import numpy as np
    rng = np.random.default_rng(4)
    queries = S["queries"] + [[0] + rng.integers(4, vocab - 8, size=int(rng.integers(4, 9))).tolist() + [2] for _ in range(2)]
    return s, queries


def _search_worker(rank, world, port, k, q):
    from seal_amd.distributed import sharded_batch_search
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, queries = _cpu_searcher(batch_size=1)
    full = sharded_batch_search(s, queries, k=k)
    q.put((rank, full.numpy().tobytes(), tuple(full.shape)))       # by value: the parent reads it after this process is gone
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_equals_the_single_process_search(monkeypatch):
    from seal_amd.distributed import pack_topk
    world, k = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_search_worker, args=(r, world, port, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    s, queries = _cpu_searcher(batch_size=1, patch=monkeypatch)           # one query per batch on both sides: identical arithmetic
    want = pack_topk(s.batch_search(queries, k=k, detokenize=False), k)
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert want.shape == (len(queries), k, 2) and (want[:, 0, 0] >= 0).all()
    import numpy as np
    for _, raw, shape in got:
        full = torch.from_numpy(np.frombuffer(raw, dtype=np.float64).reshape(shape).copy())
        assert torch.equal(full, want)          # same documents, bit-equal float64 scores, in query order, on every rank
