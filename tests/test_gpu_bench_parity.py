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
    monkeypatch.setattr(retrieval, "TITLE_MAX_LENGTH", 8)      # the searcher's default path: body + title decodes as one loop
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


def test_suffix_array_audit_passes_and_catches_a_missorted_suffix_array():
    """bench.py's builder-independent audit (adjacent suffixes compared symbol by symbol on the device, SA checksum, one
    backward-search step per sampled row landing on SA[i]-1) on a GPU-built index, then on the same index with two
    suffix-array entries swapped in place"""
    import bench
    from seal_amd import FMIndex
    dev = torch.device("cuda:0")
    data, beg, _, _ = bench.synth_corpus(400, dev, seed=0, phrases=300)
    ix = FMIndex()
    ix.initialize_from_device(data, beg.tolist())
    rep = bench.sa_audit(ix, n_pairs=1 << 17)
    assert rep["mismatches"] == 0 and rep["sum_of_sa_is_n_choose_2"] and rep["longest_common_prefix_seen"] >= 2, rep
    sa = bench.device_array(ix, "sa_lo", "<i4")
    i = ix.size() // 2
    a, b = int(sa[i]), int(sa[i + 1])
    sa[i], sa[i + 1] = b, a
    torch.cuda.synchronize()
    bad = bench.sa_audit(ix, n_pairs=1 << 19)
    assert bad["suffix_order_violations"] >= 1 and bad["bwt_lf_violations"] >= 1 and bad["sum_of_sa_is_n_choose_2"], bad
    sa[i] = a                                        # a duplicated entry: the checksum must notice
    torch.cuda.synchronize()
    assert not bench.sa_audit(ix, n_pairs=1 << 12)["sum_of_sa_is_n_choose_2"]
