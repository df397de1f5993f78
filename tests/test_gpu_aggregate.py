"""GPU evidence aggregation (seal_amd/csrc/fmi_aggregate.hip: first stage + full-document scoring of reference
seal/keys.py:311-497 on the device) against (1) the vectors the reference's own ``aggregate_evidence`` produced,
(2) the bit-exact host routines fmi_first_stage / fmi_full_score on corpora built to stress the order-sensitive
parts: overlapping windows of one key and of nested keys, tied scores, repeated tokens, every option."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLDEN, "ref_aggregate_evidence.json")) as f:
    AGG = json.load(f)["cases"]


def _unhex(x):
    return float.fromhex(x)


def _as_golden(results):
    return [{"doc": int(doc), "score": float(info[0]).hex(),
             "keys": [[[int(t) for t in k], float(s).hex()] for k, s in info[1]],
             "tokens": [int(t) for t in info[3]],
             "best": [[int(t) for t in info[4][0]], float(info[4][1]).hex()]} for doc, info in results.items()]


def _hip_index(docs):
    from seal_amd import FMIndex
    ix = FMIndex()
    ix.initialize(docs)
    return ix


@pytest.mark.parametrize("case_no", range(len(AGG)))
def test_gpu_aggregation_equals_the_reference(case_no, monkeypatch):
    """the 36 configurations the reference's own aggregate_evidence was run on: ranking, float64 scores (hex), accepted
    keys, document tokens, best key -- from the device pipeline"""
    from seal_amd import gpu_aggregate
    from seal_amd.keys import aggregate_evidence
    case = AGG[case_no]
    kw = case["kwargs"]
    ix = _hip_index(case["docs"])
    keys = [(list(k), _unhex(s)) for k, s in case["keys"]]
    us = None if case["unigram_scores"] is None else [_unhex(x) for x in case["unigram_scores"]]
    calls = []
    real = gpu_aggregate._run_plan
    monkeypatch.setattr(gpu_aggregate, "_run_plan", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    results, all_ngrams = aggregate_evidence(keys, unigram_scores=us, index=ix, **kw)
    on_gpu = not (kw.get("sort_by_length") or kw.get("sort_by_freq") or kw.get("first_stage_only"))
    if on_gpu and len(case["results"]) > 0:
        assert calls, "the device path did not run"
    assert [[list(k), float(s).hex()] for k, s in all_ngrams.items()] == [[list(k), s] for k, s in case["all_ngrams"]]
    got = _as_golden(results)
    assert [g["doc"] for g in got] == [w["doc"] for w in case["results"]]
    assert got == case["results"]


def _same(a, b, keep=None):
    (ra, na), (rb, nb) = a, b
    assert list(na.items()) == list(nb.items())
    ka, kb = list(ra.keys()), list(rb.keys())
    if keep is not None:
        kb = kb[:keep]
    assert ka == kb
    for d in ka:
        assert ra[d][0] == rb[d][0], d                                  # float64, same operation order -> exact
        assert [(tuple(n), s) for n, s in ra[d][1]] == [(tuple(n), s) for n, s in rb[d][1]], d
        assert ra[d][4][1] == rb[d][4][1] and tuple(ra[d][4][0]) == tuple(rb[d][4][0]), d
        assert list(ra[d][3]) == list(rb[d][3]), d


def _stress_corpus(seed, vocab):
    """documents with repeated spans (shared passages), runs of one token and short periods: windows of the same key
    overlap each other, nested keys end on the same position, many documents tie"""
    from tests.helpers import make_docs
    rng = np.random.default_rng(seed)
    docs = make_docs(seed, 150, vocab, min_len=6, max_len=24, title_sep=7)
    spans = [rng.integers(4, vocab, size=int(rng.integers(3, 9))).tolist() for _ in range(12)]
    for _ in range(120):
        d = []
        for _ in range(int(rng.integers(1, 5))):
            d += spans[int(rng.integers(len(spans)))] if rng.random() < 0.7 else rng.integers(4, vocab, size=3).tolist()
        docs.append(d[:2] + [7] + d[2:] + [2])
    docs.append([9] * 11 + [2])
    docs.append([9, 8] * 7 + [2])
    docs.append([9, 9, 8] * 5 + [9, 2])
    return docs, spans


def _stress_keys(rng, docs, spans, vocab):
    from tests.helpers import synthetic_keys
    keys = synthetic_keys(rng, docs, vocab, n_keys=60, with_titles=True)
    for sp in spans[:8]:                                      # nested prefixes and suffixes of shared passages, tied scores
        for n in range(1, len(sp) + 1):
            keys.append((sp[:n], -float(int(rng.integers(1, 4)))))
        keys.append((sp[1:], -2.0))
    keys += [([9, 9], -0.7), ([9, 9, 9], -1.1), ([9], -1.5), ([9, 8], -0.7), ([8, 9], -0.7), ([9, 8, 9], -1.1), ([9, 9, 8, 9], -0.9)]
    seen, out = set(), []
    for k, s in keys:
        if tuple(k) not in seen:
            seen.add(tuple(k))
            out.append((k, s))
    return out


OPTIONS = [
    dict(),
    dict(add_best_unigrams_to_ngrams=True, use_top_k_unigrams=30, n_docs_complete_score=20),
    dict(max_occurrences_1=5, n_docs_complete_score=7, beta=0.5, alpha=1.5),
    dict(single_key=0.3, allow_overlaps=True, length_penalty=0.1),
    dict(single_key=0.6, single_key_add_unigrams=True, unigrams_ignore_free_places=True),
    dict(beta=0.0), dict(beta=1.0, n_docs_complete_score=1500),
    dict(use_fm_index_frequency=False, allow_overlaps=True),
]


@pytest.mark.parametrize("python_scoring", [False, True], ids=["cpp-scoring", "python-scoring"])
@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("kw", OPTIONS, ids=[str(i) for i in range(len(OPTIONS))])
def test_gpu_aggregation_equals_the_host_routines(seed, kw, python_scoring, monkeypatch):
    from seal_amd.keys import aggregate_evidence, aggregate_evidence_batch
    vocab = 40 if seed == 2 else 60                           # the small alphabet: ties and overlaps everywhere
    rng = np.random.default_rng(seed)
    docs, spans = _stress_corpus(seed, vocab)
    ix = _hip_index(docs)
    jobs = []
    for q in range(4):
        keys = _stress_keys(rng, docs, spans, vocab)
        us = (-rng.random(vocab) * 8 - 0.01) if q != 2 else None
        jobs.append((keys, None if us is None else us.tolist()))
    for keep in (None, 5):
        monkeypatch.setenv("SEAL_HOST_AGGREGATE", "1")
        want = [aggregate_evidence(k, unigram_scores=u, index=ix, **kw) for k, u in jobs]
        monkeypatch.delenv("SEAL_HOST_AGGREGATE")
        ix._agg_debug = []
        got = aggregate_evidence_batch(jobs, ix, keep=keep, python_scoring=python_scoring,
                                       **{**dict(n_docs_complete_score=500, max_occurrences_1=1500), **kw})
        assert len(ix._agg_debug) == 1, "the device path did not run"
        fs = ix._agg_debug[0]
        ix._agg_debug = None
        assert sum(len(w[0]) for w in want) > 20
        for g, w in zip(got, want):
            _same(g, w, keep)
        # two phases (the overlapped search puts the next batch's rescoring enqueue between them): the same results
        fetch = aggregate_evidence_batch(jobs, ix, keep=keep, python_scoring=python_scoring, two_phase=True,
                                         **{**dict(n_docs_complete_score=500, max_occurrences_1=1500), **kw})
        assert callable(fetch)
        for g, w in zip(fetch(), want):
            _same(g, w, keep)
        # the GPU path's result object answers what callers ask of the host path's dict (Mapping contract, gpu_aggregate._Results)
        from collections.abc import Mapping
        from itertools import islice
        res, host = got[0][0], want[0][0]
        res = res.result() if hasattr(res, "result") else res
        assert isinstance(res, Mapping) and isinstance(host, Mapping) and hasattr(res, "to_dict"), type(res)
        n = len(res)
        assert list(res) == list(res.keys()) == list(host)[:n] and len(res.items()) == n and len(list(res.values())) == n
        first = next(iter(res))
        assert first in res and res[first][0] == host[first][0] and list(islice(res.items(), 2)) == list(res.items())[:2]
        as_dict = res.to_dict()
        assert type(as_dict) is dict and list(as_dict) == list(res) and as_dict[first][0] == host[first][0] and as_dict[first][3] == list(host[first][3])
        # the first-stage ranking itself (keys.py:366) against the host routine
        monkeypatch.setenv("SEAL_HOST_AGGREGATE", "1")
        for (fd, fsc), (k, u) in zip(fs, jobs):
            ranked, _ = aggregate_evidence(k, unigram_scores=u, index=ix, first_stage_only=True, **kw)
            assert fd.tolist() == list(ranked.keys())
            assert fsc.tolist() == [info[0] for info in ranked.values()]
        monkeypatch.delenv("SEAL_HOST_AGGREGATE")


def test_gpu_aggregation_long_documents_and_overflowing_candidate_lists():
    """documents of hundreds of tokens over a 6-symbol alphabet with every short n-gram as a key: thousands of key
    occurrences per document (beyond the LDS candidate list -> global pool), long clusters of overlapping windows"""
    from seal_amd.keys import aggregate_evidence
    import os as _os
    rng = np.random.default_rng(7)
    docs = [rng.integers(4, 10, size=int(rng.integers(200, 500))).tolist() + [2] for _ in range(12)]
    docs += [rng.integers(4, 10, size=int(rng.integers(5, 30))).tolist() + [2] for _ in range(60)]
    ix = _hip_index(docs)
    keys, seen = [], set()
    for n in (1, 2, 3, 4):
        for _ in range(120):
            k = tuple(rng.integers(4, 10, size=n).tolist())
            if k not in seen:
                seen.add(k)
                keys.append((list(k), -float(rng.random() * 5 + 0.05)))
    us = (-rng.random(16) * 6 - 0.01).tolist()
    kw = dict(max_occurrences_1=100000, n_docs_complete_score=100)
    _os.environ["SEAL_HOST_AGGREGATE"] = "1"
    try:
        want = aggregate_evidence(keys, unigram_scores=us, index=ix, **kw)
    finally:
        del _os.environ["SEAL_HOST_AGGREGATE"]
    got = aggregate_evidence(keys, unigram_scores=us, index=ix, **kw)
    assert len(want[0]) == 72
    _same(got, want)


@pytest.mark.parametrize("nbytes", [16, 4096, 3_000_016, 20, 1_000_004])
def test_kernel_copy_moves_bytes_both_ways_and_refuses_pageable_memory(nbytes):
    """``fmi_dev_kernel_copy`` (the aggregation's plan goes up and its records come back through it, not through the DMA queue every
    stream of the process shares): pinned -> device and device -> pinned, 16-byte and 4-byte granular, by value"""
    import torch
    from seal_amd._lib import SealFMError, check, lib
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev, priority=-1)
    src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8).pin_memory()
    mid = torch.zeros(nbytes + 64, dtype=torch.uint8, device=dev)
    back = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
    with torch.cuda.stream(st):
        check(lib().fmi_dev_kernel_copy(st.cuda_stream, mid.data_ptr(), src.data_ptr(), nbytes))
        check(lib().fmi_dev_kernel_copy(st.cuda_stream, back.data_ptr(), mid.data_ptr(), nbytes))
    st.synchronize()
    assert torch.equal(back, src)
    assert torch.equal(mid[:nbytes].cpu(), src) and int(mid[nbytes:].sum()) == 0
    if nbytes == 4096:
        pageable = torch.zeros(nbytes, dtype=torch.uint8)
        with pytest.raises(SealFMError):
            check(lib().fmi_dev_kernel_copy(st.cuda_stream, mid.data_ptr(), pageable.data_ptr(), nbytes))
        with pytest.raises(SealFMError):
            check(lib().fmi_dev_kernel_copy(st.cuda_stream, mid.data_ptr() + 2, src.data_ptr(), 8))


@pytest.mark.parametrize("n_docs,n_top", [(3000, 1), (3000, 100), (3000, 257), (3000, 2999), (3000, 3000), (3000, 4000), (15000, 6000)])
def test_first_stage_ranking_by_selection_equals_the_three_stable_sorts_under_mass_ties(n_docs, n_top):
    """``k_select_top`` (round 6: per query the n_docs_complete_score best entries by an MSD radix select on (rank key, first touch)) against the
    three full stable sorts of rounds 2-5 (``agg_rank_by_sorts``) and against the host routine, on the case the selection has to work hardest
    for: thousands of documents with the SAME score (one occurrence of one key each), the cut falling inside the tie group -- the order among
    them is the order of first touch (Python's stable sort over the dict's insertion order, keys.py:366-375) --, plus groups above and below.
    (15 000 documents, 6 000 wanted: the cut falls into a tie group of 10 000, more than k_sel_final can hold beside 6 000 selected entries -- the
    query takes the single-workgroup fallback inside the default pipeline.)"""
    from seal_amd.keys import aggregate_evidence
    from tests.helpers import kernel_options
    rng = np.random.default_rng(11)
    docs = []
    for d in range(n_docs):
        filler = rng.integers(20, 60, size=int(rng.integers(2, 6))).tolist()
        docs.append(filler[:1] + [7] + filler[1:] + [10 + (d % 3 == 0)] + [2])          # token 10 or 11 once per document
    ix = _hip_index(docs)
    keys = [([10], -1.0), ([11], -1.0), ([11, 2], -0.5)]                              # [11, 2]: the same documents as [11], a better score
    keys += [(docs[d][:2], -0.25) for d in (5, 17, 1500)]                               # a few documents on top
    kw = dict(max_occurrences_1=20000, n_docs_complete_score=n_top, use_fm_index_frequency=False)
    os.environ["SEAL_HOST_AGGREGATE"] = "1"
    try:
        want = aggregate_evidence(keys, unigram_scores=None, index=ix, **kw)
    finally:
        del os.environ["SEAL_HOST_AGGREGATE"]
    by_selection = aggregate_evidence(keys, unigram_scores=None, index=ix, **kw)
    with kernel_options(ix, agg_rank_by_sorts=1):
        by_sorts = aggregate_evidence(keys, unigram_scores=None, index=ix, **kw)
    with kernel_options(ix, agg_rank_by_sorts=2):          # the single-workgroup form (what a tie group beyond the `maybe` buffer falls back to)
        by_one_workgroup = aggregate_evidence(keys, unigram_scores=None, index=ix, **kw)
    assert len(want[0]) == min(n_top, n_docs)
    _same(by_selection, want)
    _same(by_sorts, want)
    _same(by_one_workgroup, want)
