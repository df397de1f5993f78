"""Host logic of seal_amd.keys on CPU: the batched/vectorised aggregate_evidence
against the scalar, stage-by-stage model of the reference (oracle/keys_oracle.py)."""
import numpy as np
import pytest

from oracle.keys_oracle import oracle_aggregate_evidence, oracle_deduplicate, oracle_strip
from oracle.seal_oracle import OracleFMIndex
from seal_amd.keys import aggregate_evidence, deduplicate, strip
from tests.helpers import OracleBatchIndex, make_docs, synthetic_keys


def _same(a, b):
    (ra, na), (rb, nb) = a, b
    assert list(na.items()) == list(nb.items())
    assert list(ra.keys()) == list(rb.keys())
    for d in ra:
        assert ra[d][0] == rb[d][0], d                      # float64, same operation order -> exact
        assert [(tuple(n), s) for n, s in ra[d][1]] == [(tuple(n), s) for n, s in rb[d][1]]
        assert ra[d][-1][1] == rb[d][-1][1] and tuple(ra[d][-1][0]) == tuple(rb[d][-1][0])
        if len(ra[d]) == 5:
            assert ra[d][3] == rb[d][3]


@pytest.mark.parametrize("seed,kw", [
    (0, dict()),
    (1, dict(add_best_unigrams_to_ngrams=True, use_top_k_unigrams=30, n_docs_complete_score=20)),
    (2, dict(max_occurrences_1=5, n_docs_complete_score=7, beta=0.5, alpha=1.5)),
    (3, dict(first_stage_only=True, max_occurrences_1=50)),
    (4, dict(single_key=0.3, allow_overlaps=True, length_penalty=0.1)),
    (5, dict(sort_by_length=True)),
    (6, dict(use_fm_index_frequency=False)),
])
def test_aggregate_evidence_matches_restatement(seed, kw):
    vocab = 60
    rng = np.random.default_rng(seed)
    docs = make_docs(seed, 120, vocab, min_len=6, max_len=20, title_sep=7)
    # a repetitive document so that windows of one key overlap each other
    docs.append([9, 9, 9, 9, 9, 9, 9, 9, 2])
    orc = OracleFMIndex()
    orc.initialize(docs)
    keys = synthetic_keys(rng, docs, vocab, with_titles=True) + [([9, 9], -0.7), ([9, 9, 9], -1.1)]
    us = (-rng.random(vocab) * 8 - 0.01).tolist()
    got = aggregate_evidence(keys, unigram_scores=us, index=OracleBatchIndex(orc), **kw)
    want = oracle_aggregate_evidence(keys, unigram_scores=us, index=orc, **kw)
    assert len(want[0]) > 0
    _same(got, want)
    got = aggregate_evidence(keys, unigram_scores=None, index=OracleBatchIndex(orc), **kw)
    want = oracle_aggregate_evidence(keys, unigram_scores=None, index=orc, **kw)
    _same(got, want)


def test_small_helpers():
    assert strip([2, 0, 5, 6, 2], (0, 2), (2,)) == oracle_strip([2, 0, 5, 6, 2], (0, 2), (2,)) == [5, 6]
    assert strip([2, 2], (2,), (2,)) == oracle_strip([2, 2], (2,), (2,)) == []
    items = [(0.5, [1, 2]), (0.4, [1, 2]), (0.1, [3])]
    assert deduplicate(items) == oracle_deduplicate(items) == [(0.5, [1, 2]), (0.1, [3])]
    assert deduplicate([[1], [1], [2]]) == [[1], [2]]


@pytest.mark.parametrize("tree", ["1", "0"], ids=["prefix-tree", "maximal-parent-rows"])
def test_rescore_keys_prefix_sharing_matches_row_per_key(tree, monkeypatch):
    """both prefix-sharing forwards -- one decoder position per DISTINCT prefix (tree attention, the default) and one row
    per maximal parent -- against the reference's one row per key (keys.py:64-141)"""
    import torch
    from seal_amd.keys import rescore_keys
    from seal_amd import keys as keys_mod
    monkeypatch.setattr(keys_mod, "RESCORE_TREE", tree == "1")
    monkeypatch.setattr(keys_mod, "RESCORE_MAX_NODES", 20)          # several forwards per call: chunks of whole queries
    from tests.helpers import tiny_bart
    m = tiny_bart(120)
    rng = np.random.default_rng(0)
    inputs = [[0] + rng.integers(4, 118, size=6).tolist() + [2] for _ in range(3)]
    keys = []
    for _ in range(3):
        base = rng.integers(4, 118, size=7).tolist()
        other = rng.integers(4, 118, size=5).tolist()
        kk = [base[:i] for i in range(1, 8)] + [other[:i] for i in range(2, 6)] + [[2] + base[:3], base[:4] + [2], [7] + other[:2] + [7]]
        keys.append([(-1.0, k) for k in kk])
    for kw in (dict(), dict(strip_from_bos=[2, 7], strip_from_eos=[7, 2]), dict(prefix=[5]), dict(length_penalty=1.0)):
        a = rescore_keys(m, inputs, keys, batch_size=4, share_prefixes=True, **kw)
        b = rescore_keys(m, inputs, keys, batch_size=4, share_prefixes=False, **kw)
        for qa, qb in zip(a, b):
            assert [k for _, k in qa] == [k for _, k in qb]
            for (sa, _), (sb, _) in zip(qa, qb):
                assert abs(sa - sb) <= 2e-5 * max(1.0, abs(sb)), kw


def test_deferred_first_stage_in_a_worker_process_matches_inline():
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    vocab = 60
    rng = np.random.default_rng(3)
    docs = make_docs(3, 120, vocab, min_len=6, max_len=20, title_sep=7)
    orc = OracleFMIndex()
    orc.initialize(docs)
    keys = synthetic_keys(rng, docs, vocab, with_titles=True)
    us = (-rng.random(vocab) * 8 - 0.01).tolist()
    kw = dict(unigram_scores=us, index=OracleBatchIndex(orc), first_stage_only=True, add_best_unigrams_to_ngrams=True,
              use_top_k_unigrams=30)
    inline, ng1 = aggregate_evidence(keys, **kw)
    with ProcessPoolExecutor(max_workers=1, mp_context=multiprocessing.get_context("spawn")) as pool:
        handle, ng2 = aggregate_evidence(keys, defer=pool, keep=7, **kw)
        got = handle.result()
    assert list(ng1.items()) == list(ng2.items())
    want = list(inline.items())[:7]
    assert [d for d, _ in want] == list(got.keys())
    for d, info in want:
        assert got[d][0] == info[0] and [(tuple(n), s) for n, s in got[d][1]] == [(tuple(n), s) for n, s in info[1]]
        assert tuple(got[d][2][0]) == tuple(info[2][0]) and got[d][2][1] == info[2][1]


@pytest.mark.parametrize("first_stage_only", [True, False])
def test_aggregate_evidence_batch_equals_per_query_calls(first_stage_only):
    from seal_amd.keys import aggregate_evidence_batch
    vocab = 60
    rng = np.random.default_rng(11)
    docs = make_docs(11, 150, vocab, min_len=6, max_len=20, title_sep=7)
    orc = OracleFMIndex()
    orc.initialize(docs)
    jobs = []
    for q in range(4):
        keys = synthetic_keys(rng, docs, vocab, with_titles=True)
        us = (-rng.random(vocab) * 8 - 0.01) if q != 2 else None
        jobs.append((keys, us))
    jobs.append(([], None))      # a query without keys
    params = dict(first_stage_only=first_stage_only, add_best_unigrams_to_ngrams=True, use_top_k_unigrams=25,
                  n_docs_complete_score=30, max_occurrences_1=40)
    got = aggregate_evidence_batch(jobs, OracleBatchIndex(orc), **params)
    later = aggregate_evidence_batch(jobs, OracleBatchIndex(orc), two_phase=True, **params)       # (host route: the callable does all the work)
    assert callable(later)
    for (keys, us), g, g2 in zip(jobs, got, later()):
        want = oracle_aggregate_evidence(keys, unigram_scores=None if us is None else us.tolist(), index=orc, **params)
        _same(g, want)
        _same(g2, want)


@pytest.mark.parametrize("seed", range(12))
def test_full_scoring_native_stress_small_alphabet(seed):
    """tiny alphabet + quantised scores: nested / overlapping / tied keys everywhere, to exercise the
    registration order (odd-ascending, even-descending lengths), heap tie-breaks and coverage discounts"""
    vocab = 9
    rng = np.random.default_rng(100 + seed)
    docs = [rng.integers(3, vocab, size=int(rng.integers(4, 30))).tolist() + [2] for _ in range(40)]
    orc = OracleFMIndex()
    orc.initialize(docs)
    keys = []
    for _ in range(60):
        d = docs[int(rng.integers(len(docs)))]
        a = int(rng.integers(0, len(d) - 1))
        ng = d[a:a + int(rng.integers(1, 7))]
        keys.append((list(ng), -float(rng.integers(1, 6)) / 2.0))          # few distinct scores -> ties
    us = (-rng.integers(1, 9, size=vocab) / 2.0).tolist()
    kw = dict(n_docs_complete_score=25, max_occurrences_1=int(rng.integers(3, 200)), add_best_unigrams_to_ngrams=bool(seed % 2),
              use_top_k_unigrams=6, single_key=float(seed % 3) / 4.0, allow_overlaps=bool(seed % 5 == 0),
              single_key_add_unigrams=bool(seed % 4 == 1), unigrams_ignore_free_places=bool(seed % 7 == 3))
    got = aggregate_evidence(keys, unigram_scores=us, index=OracleBatchIndex(orc), **kw)
    want = oracle_aggregate_evidence(keys, unigram_scores=us, index=orc, **kw)
    assert len(want[0]) > 0
    _same(got, want)


def test_deferred_full_scoring_in_a_worker_process_matches_inline():
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    vocab = 60
    rng = np.random.default_rng(4)
    docs = make_docs(4, 120, vocab, min_len=6, max_len=20, title_sep=7)
    orc = OracleFMIndex()
    orc.initialize(docs)
    keys = synthetic_keys(rng, docs, vocab, with_titles=True)
    us = (-rng.random(vocab) * 8 - 0.01).tolist()
    kw = dict(unigram_scores=us, index=OracleBatchIndex(orc), add_best_unigrams_to_ngrams=True, use_top_k_unigrams=30,
              n_docs_complete_score=40)
    inline, _ = aggregate_evidence(keys, **kw)
    with ProcessPoolExecutor(max_workers=1, mp_context=multiprocessing.get_context("spawn")) as pool:
        handle, _ = aggregate_evidence(keys, defer=pool, keep=9, **kw)
        got = handle.result()
    want = list(inline.items())[:9]
    assert [d for d, _ in want] == list(got.keys())
    for d, info in want:
        assert got[d][0] == info[0] and got[d][3] == info[3]
        assert [(tuple(n), s) for n, s in got[d][1]] == [(tuple(n), s) for n, s in info[1]]
        assert tuple(got[d][4][0]) == tuple(info[4][0]) and got[d][4][1] == info[4][1]


def test_aggregate_evidence_random_option_sweep():
    """60 random draws over every option of aggregate_evidence (both stages, all three "best key"
    modes, overlaps, single-key blending, unigram fill variants, tied unigram scores, repeated keys,
    tight occurrence / shortlist limits): product == oracle, exactly"""
    import random
    opts = dict(
        with_unigrams=[False, True], first_stage_only=[False, True], mode=["score", "length", "freq"],
        allow_overlaps=[False, True], single_key=[0.0, 0.3, 1.0], use_fm_index_frequency=[True, False],
        add_best_unigrams_to_ngrams=[False, True], single_key_add_unigrams=[False, True],
        unigrams_ignore_free_places=[False, True], max_occurrences_1=[3, 20, 1500], n_docs_complete_score=[2, 10, 500],
        use_top_k_unigrams=[0, 5, 1000], beta=[0.0, 0.8, 1.0], alpha=[1.0, 2.0], length_penalty=[0.0, 0.2], smoothing=[5.0, 0.5])
    pick = random.Random(7)
    compared = 0
    for it in range(60):
        kw = {k: pick.choice(v) for k, v in opts.items()}
        mode = kw.pop("mode")
        kw["sort_by_length"], kw["sort_by_freq"] = mode == "length", mode == "freq"
        with_unigrams = kw.pop("with_unigrams")
        vocab, n_docs = pick.choice([12, 40, 300]), pick.choice([5, 40, 150])
        rng = np.random.default_rng(1000 + it)
        docs = make_docs(1000 + it, n_docs, vocab, title_sep=7)
        orc = OracleFMIndex()
        orc.initialize(docs)
        keys = synthetic_keys(rng, docs, vocab, n_keys=int(rng.integers(5, 50)), with_titles=True)
        if it % 2:
            keys += [keys[0], (keys[1][0], keys[1][1] - 0.3)]       # a repeated key keeps its place, takes the later score
        us = None
        if with_unigrams:
            us = (-(rng.random(vocab + 12) * 9 + 0.05)).tolist()
            if it % 3 == 0:
                us = [float(min(round(x), -1)) for x in us]        # ties among unigram scores
        got = aggregate_evidence(keys, unigram_scores=us, index=OracleBatchIndex(orc), **kw)
        want = oracle_aggregate_evidence(keys, unigram_scores=us, index=orc, **kw)
        _same(got, want)
        compared += len(want[0])
    assert compared > 300


@pytest.mark.parametrize("npre", [0, 1])
def test_prefix_tree_equals_a_dictionary_built_forest(npre):
    """keys._prefix_tree (numpy, level by level) against the obvious construction: one node per distinct (query, prefix),
    every term (key, position >= npre) pointing at the node of its prefix; ancestors root first, the node itself last"""
    from seal_amd.keys import _prefix_tree
    rng = np.random.default_rng(5 + npre)
    key_query, key_seqs = [], []
    for q in (0, 3, 4):                                     # query ids need not be dense
        stems = [tuple(rng.integers(2, 9, size=rng.integers(1, 6)).tolist()) for _ in range(6)]
        for _ in range(40):
            s = stems[rng.integers(len(stems))]
            k = s[:rng.integers(1, len(s) + 1)] + tuple(rng.integers(0, 9, size=rng.integers(0, 3)).tolist())
            if len(k) > npre:
                key_query.append(q); key_seqs.append(k)
    key_query.append(3); key_seqs.append(key_seqs[0] if key_query[0] == 3 else key_seqs[-1])       # a duplicate key
    t = _prefix_tree(key_query, key_seqs, npre, start=99)
    n = len(t["tok"])
    want = {(q, k[:j]) for q, k in zip(key_query, key_seqs) for j in range(len(k))}
    assert n == len(want) and t["anc"].shape == (n, max(len(k) for k in key_seqs))
    prefix_of = {}
    for i in range(n):
        d = int(t["depth"][i])
        row = t["anc"][i]
        assert row[d] == i and (row[d + 1:] == -1).all() and (row[:d + 1] >= 0).all()
        toks = tuple(int(t["tok"][a]) for a in row[1:d + 1])
        assert int(t["tok"][row[0]]) == 99 and all(int(t["depth"][a]) == x for x, a in enumerate(row[:d + 1]))
        assert all(int(t["query"][a]) == int(t["query"][i]) for a in row[:d + 1])
        prefix_of[i] = (int(t["query"][i]), toks)
    assert set(prefix_of.values()) == want and len(set(prefix_of.values())) == n
    terms = sorted(zip(t["term_key"].tolist(), t["term_col"].tolist(), t["term_node"].tolist(), t["term_tok"].tolist()))
    exp = sorted((k, j - npre, None, key_seqs[k][j]) for k in range(len(key_seqs)) for j in range(npre, len(key_seqs[k])))
    assert len(terms) == len(exp)
    for (k, c, node, tok), (ek, ec, _, etok) in zip(terms, exp):
        assert (k, c, tok) == (ek, ec, etok)
        assert prefix_of[node] == (key_query[k], key_seqs[k][:c + npre])
    assert t["width"] == max(1, max(len(k) for k in key_seqs) - npre)


def test_rescore_keys_multi_equals_separate_calls():
    """jobs with different encoder inputs (and lengths), prefixes and strip lists through ONE forest / one forward: each
    job's scores equal its own rescore_keys call (and the reference's one row per key)"""
    import torch
    from seal_amd.keys import rescore_keys, rescore_keys_multi
    from tests.helpers import tiny_bart
    m = tiny_bart(120)
    rng = np.random.default_rng(1)

    def some_keys(nq):
        out = []
        for _ in range(nq):
            base = rng.integers(4, 118, size=6).tolist()
            other = rng.integers(4, 118, size=4).tolist()
            kk = [base[:i] for i in range(1, 7)] + [other[:i] for i in range(1, 5)] + [[2] + base[:2], base[:3] + [2]]
            out.append([(-1.0, k) for k in kk])
        return out
    jobs = [
        (m, [[0] + rng.integers(4, 118, size=5).tolist() + [2] for _ in range(3)], some_keys(3), dict(strip_from_bos=[2], strip_from_eos=[2])),
        (m, [[0] + rng.integers(4, 118, size=9).tolist() + [2] for _ in range(3)], some_keys(3), dict(prefix=[5])),
        (m, [[0] + rng.integers(4, 118, size=2).tolist() + [2] for _ in range(2)], some_keys(2), dict(logit_bias=torch.randn(2, 120))),
    ]
    multi = rescore_keys_multi(jobs)
    for (model, inputs, keys, kw), got in zip(jobs, multi):
        for share in (True, False):
            want = rescore_keys(model, inputs, keys, batch_size=4, share_prefixes=share, **kw)
            for qa, qb in zip(got, want):
                assert [k for _, k in qa] == [k for _, k in qb]
                for (sa, _), (sb, _) in zip(qa, qb):
                    assert abs(sa - sb) <= 2e-5 * max(1.0, abs(sb)), (kw, share)


def test_array_post_filters_equal_the_reference_list_comprehensions():
    """``retrieval._HypArrays`` (the searcher's post-filters on the decode history as arrays) against the reference's list
    comprehensions (retrieval.py:85-90 body keys, 180-190 title keys) on random hypotheses that hit every edge: keys that
    become empty after one / two / three strips, keys of strip tokens only, invalid (-inf) scores, the length filters,
    titles that do not end with the title eos, titles that do not start with the title bos."""
    import numpy as np
    from seal_amd.retrieval import _HypArrays
    rng = np.random.default_rng(0)
    strip_ids, title_eos, title_bos = (0, 2), 7, 2
    B, H, L = 5, 300, 6
    length = rng.integers(1, L + 1, size=H)
    tok = np.full((B, H, L), -1, dtype=np.int64)
    for b in range(B):
        for h in range(H):
            tok[b, h, :length[h]] = rng.choice([0, 2, 2, 7, 7, 5, 9, 11], size=length[h])
    score = rng.normal(size=(B, H))
    valid = rng.random((B, H)) > 0.1

    def lists():
        return [[(float(score[b, h]), tok[b, h, :length[h]].tolist()) for h in range(H) if valid[b, h]] for b in range(B)]

    for min_length in (0, 2):
        # body keys
        want = lists()
        for fk in want:
            fk[:] = [(sc, k[1:] if k[0] in strip_ids else k) for sc, k in fk if k]
            fk[:] = [(sc, k[1:] if k[0] in strip_ids else k) for sc, k in fk if k]
            fk[:] = [(sc, k[:-1] if k[-1] in strip_ids else k) for sc, k in fk if k]
            if min_length > 0:
                fk[:] = [(sc, k) for sc, k in fk if len(k) == min_length]
            fk[:] = [(sc, k) for sc, k in fk if k]            # what `if k and count > 0` keeps before counting
        arr = _HypArrays((tok, length, score, valid))
        arr.drop_empty(); arr.strip_front(strip_ids)
        arr.drop_empty(); arr.strip_front(strip_ids)
        arr.drop_empty(); arr.strip_back(strip_ids)
        if min_length > 0:
            arr.require_length(min_length)
        idx, lens, flat = arr.csr()
        got = arr.lists(idx)
        assert got == want
        assert lens.tolist() == [len(k) for fk in want for _, k in fk] and flat.tolist() == [t for fk in want for _, k in fk for t in k]
        # title keys (every hypothesis has at least one token, as every recorded hypothesis does)
        want = lists()
        for fk in want:
            fk[:] = [(sc, k[:-1] if k[-1] in strip_ids else k) for sc, k in fk]
            fk[:] = [(sc, k) for sc, k in fk if k and k[-1] == title_eos]
            if min_length > 0:
                fk[:] = [(sc, k) for sc, k in fk if len(k) == (min_length + 1)]
            fk[:] = [(sc, [title_bos] + k if k[0] != title_bos else k) for sc, k in fk]
        arr = _HypArrays((tok, length, score, valid))
        arr.strip_back(strip_ids)
        arr.require_last(title_eos)
        if min_length > 0:
            arr.require_length(min_length + 1)
        idx, _, _ = arr.csr()
        assert arr.lists(idx, prepend=title_bos) == want
