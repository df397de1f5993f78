"""bench.py's own parity machinery at test scale: the operations a searcher batch issues are recorded with the
GPU's answers, replayed on the CPU oracle rebuilt from the index's device arrays (as bench.py does at NQ scale),
and compared bit for bit; a corrupted answer must be caught."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_bench_parity_check_passes_on_a_real_batch_and_catches_a_corrupted_answer(monkeypatch):
    import bench
    from seal_amd import FMIndex, retrieval
    from seal_amd.retrieval import SEALSearcher
    from tests.helpers import make_docs, tiny_bart
    vocab, title_eos = 120, 7
    dev = torch.device("cuda:0")
    docs = make_docs(5, 300, vocab - 8, min_len=6, max_len=18, title_sep=title_eos)
    ix = FMIndex()
    ix.initialize(docs)
    real = retrieval.fm_index_generate
    monkeypatch.setattr(retrieval, "fm_index_generate",
                        lambda *a, **kw: real(*a, **{**kw, "max_length": 8 if kw.get("force_decoding_from") else kw["max_length"]}))
    s = SEALSearcher(ix, None, tiny_bart(vocab, d_model=128, heads=2).to(dev), backbone="bart-tiny", length=6, beam=4, batch_size=3,
                     add_query_to_keys=False, detokenize=False, title_eos_token_id=title_eos, code_eos_token_id=vocab - 6,
                     code_bos_token_id=title_eos,
                     marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5]})
    rng = np.random.default_rng(0)
    queries = [[0] + rng.integers(4, vocab - 8, size=int(rng.integers(4, 9))).tolist() + [2] for _ in range(3)]
    ix._trace = []
    s.batch_search(queries, k=10)
    trace, ix._trace = ix._trace, None
    kinds = {op[0] for op in trace}
    assert {"mask", "ranges", "locate", "docs"} <= kinds
    orc = bench.build_cpu_oracle(ix, threads=2)
    rep, answers = bench.replay_on_cpu(orc, trace, ix.beginnings, 2, vocab=vocab)
    par = bench.parity_check(ix, trace, answers, vocab=vocab)
    assert par["mismatches"] == 0 and par["ops"] >= len(trace) and par["values_compared"] > 100
    assert all(v["values"] > 0 for v in par["by_kind"].values()) and len(par["by_kind"]) == 5
    # a single wrong located position / one flipped token bit must be reported
    i = next(j for j, op in enumerate(trace) if op[0] == "locate" and len(op[4]))
    trace[i][4][0] += 1
    j = next(j for j, op in enumerate(trace) if op[0] == "mask")
    answers[j][1][0, 0] ^= np.uint32(1 << 9)
    par = bench.parity_check(ix, trace, answers, vocab=vocab)
    assert par["by_kind"]["located_positions_and_doc_ids"]["mismatches"] == 1
    assert par["by_kind"]["allowed_token_sets"]["mismatches"] == 1
