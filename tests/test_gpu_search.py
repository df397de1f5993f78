"""GPU parity of the whole search path (keys -> evidence -> ranked docs):
SEALSearcher on cuda:0 against a scalar CPU pipeline assembled from the oracle
pieces, same seeded tiny BART and corpus.  Doc ids must match exactly."""
import numpy as np
import pytest
import torch

from tests.helpers import MODEL_GEOMETRIES, MODEL_IDS

pytestmark = pytest.mark.gpu

TITLE_EOS = 7


def _oracle_pipeline(model, orc, queries, K, length, vocab, first_stage_only, title_length=8, return_keys=False, query_keys=False,
                     decode_code=False):
    from oracle.beam_oracle import oracle_fm_index_generate
    from oracle.keys_oracle import (oracle_aggregate_evidence, oracle_body_postfilter, oracle_deduplicate,
                                    oracle_title_postfilter)
    from seal_amd import keys as rk
    from tests.helpers import hf_logits_fn
    pad = model.config.pad_token_id
    mark = {"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5], "code": [vocab - 2, vocab - 7]}

    def enc(kind):
        toks = [q[:-1] + mark[kind] + mark["+"] + q[-1:] for q in queries]
        ids = rk._pad_batch(toks, pad, "cpu")
        return toks, ids, (ids != pad).long()

    toks, ids, am = enc("body")
    body = oracle_fm_index_generate(hf_logits_fn(model, ids, am, K), orc, len(queries), K, length, vocab,
                                    pad_token_id=pad, eos_token_id=2, length_penalty=0.0)
    body = [oracle_body_postfilter(fk, orc) for fk in body]
    body = rk.rescore_keys(model, queries, body, strip_from_bos=[2, TITLE_EOS, 2], strip_from_eos=[TITLE_EOS, vocab - 6, 2])
    if query_keys:       # reference retrieval.py:115-149 for one-token words: every 1..3-gram of the query that occurs, rescored
        for q, fk, row in zip(queries, body, rk.rescore_keys(model, toks, [
                [list(k) for k in dict.fromkeys(tuple(q[1:-1][i:j]) for i in range(len(q) - 2) for j in range(i + 1, min(len(q) - 2, i + 3) + 1))
                 if orc.get_count(list(k)) > 0] for q in queries])):
            fk += row
    ttoks, ids, am = enc("title")
    title = oracle_fm_index_generate(hf_logits_fn(model, ids, am, K), orc, len(queries), K, title_length, vocab,
                                     pad_token_id=pad, eos_token_id=TITLE_EOS, length_penalty=0.0, force_decoding_from=[2])
    title = [oracle_title_postfilter(fk, orc, title_bos=2, title_eos=TITLE_EOS) for fk in title]
    title = rk.rescore_keys(model, ttoks, title, strip_from_bos=[2, TITLE_EOS, 2], strip_from_eos=[2])
    code = [[] for _ in queries]
    if decode_code:      # reference retrieval.py:212-264 with partial_code=True; code_bos = TITLE_EOS, code_eos = vocab - 6 as the searchers under test
        ctoks, ids, am = enc("code")
        code = oracle_fm_index_generate(hf_logits_fn(model, ids, am, K), orc, len(queries), K, title_length, vocab,
                                        pad_token_id=pad, eos_token_id=vocab - 6, length_penalty=0.0, force_decoding_from=[TITLE_EOS])
        strip_ids = {0, 1, 2}
        for fk in code:
            fk[:] = [(sc, k[1:-1] if k[-1] in strip_ids else k[1:]) for sc, k in fk if k]
            fk[:] = [(sc, [TITLE_EOS] + k if k[0] != TITLE_EOS else k) for sc, k in fk if k]
            fk[:] = [(sc, k) for sc, k in fk if k and orc.get_count(k) > 0]
        code = rk.rescore_keys(model, ctoks, code, strip_from_bos=[2, TITLE_EOS, 2], strip_from_eos=[2])
    out, all_keys = [], []
    uni = rk.compute_unigram_scores(model, toks)
    for b, t, c, u in zip(body, title, code, uni):
        keys = [(n, s) for s, n in oracle_deduplicate(b + t + c)]
        all_keys.append(keys)
        out.append(oracle_aggregate_evidence(keys, unigram_scores=u, index=orc, max_occurrences_1=1500,
                                             n_docs_complete_score=1500, alpha=2.0, beta=0.8, add_best_unigrams_to_ngrams=True,
                                             use_top_k_unigrams=5000, smoothing=5.0, first_stage_only=first_stage_only)[0])
    return (out, all_keys) if return_keys else out


def _tie_groups(items, rel=1e-3):
    """ranked [(doc, info)] -> per rank position, the set of documents whose scores chain to that position's
    within the fp tolerance (scores from two devices may order such documents either way)"""
    groups, cur = [], [0]
    for i in range(1, len(items)):
        a, b = items[i - 1][1][0], items[i][1][0]
        if abs(a - b) > rel * max(1.0, abs(a)):
            groups.append(cur)
            cur = []
        cur.append(i)
    groups.append(cur)
    out = [None] * len(items)
    for g in groups:
        ids = {items[i][0] for i in g}
        for i in g:
            out[i] = ids
    return out


@pytest.mark.parametrize("geom", MODEL_GEOMETRIES, ids=MODEL_IDS)
@pytest.mark.parametrize("first_stage_only,jobs,query_keys", [(False, 1, False), (True, 1, False), (False, 2, False), (True, 2, False), (False, 1, True),
                                                              (False, 1, "code")])
def test_searcher_ranks_like_the_scalar_pipeline(first_stage_only, jobs, query_keys, geom, monkeypatch):
    from oracle.seal_oracle import OracleFMIndex
    from seal_amd import FMIndex
    from seal_amd import retrieval
    from seal_amd.retrieval import SEALSearcher
    from tests.helpers import make_docs, tiny_bart
    vocab = 120
    decode_code = query_keys == "code"       # the third decode of retrieval.py:212-264 (partial_code: the corpus has no code sections)
    query_keys = False if decode_code else query_keys
    dev = torch.device("cuda:0")
    docs = make_docs(5, 200, vocab - 8, min_len=6, max_len=18, title_sep=TITLE_EOS)
    ix, orc = FMIndex(), OracleFMIndex()
    ix.initialize(docs)
    orc.initialize(docs)
    ix.labels = [f"d{i}" for i in range(len(docs))]
    rng = np.random.default_rng(0)
    queries = [[0] + rng.integers(4, vocab - 8, size=int(rng.integers(4, 9))).tolist() + [2] for _ in range(3)]
    K, length = 4, 6
    # the title decode length (15) is hard-wired in the reference; shorten it for the tiny corpus on both sides
    monkeypatch.setattr(retrieval, "TITLE_MAX_LENGTH", 8)
    s = SEALSearcher(ix, None, tiny_bart(vocab, **geom).to(dev), backbone="bart-tiny", length=length, beam=K, batch_size=2,
                     add_query_to_keys=query_keys, detokenize=False, first_stage_only=first_stage_only, jobs=jobs,
                     title_eos_token_id=TITLE_EOS, code_eos_token_id=vocab - 6, code_bos_token_id=TITLE_EOS, decode_code=decode_code,
                     partial_code=decode_code,
                     marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5],
                                       "code": [vocab - 2, vocab - 7]})
    got = s.batch_search(queries, k=10)
    st = s.bart_model._seal_step_decoder._st
    assert st.fused is (geom["d_model"] // geom["heads"] == 64)       # no silent fallback from the sealnn_* kernels
    want = _oracle_pipeline(tiny_bart(vocab, **geom), orc, queries, K, length, vocab, first_stage_only, query_keys=query_keys, decode_code=decode_code)
    for g, w in zip(got, want):
        w_all = list(w.items())
        w_items = w_all[:10]
        assert len(g) == len(w_items) > 0
        # rescored key scores come from the model on two devices -> scores within 1e-3 relative, and every
        # rank position holds a document of the scalar pipeline's tie group for that position (a group of
        # one wherever its scores are separated by more than the tolerance: then the ids match exactly)
        for d, (wd, winfo) in zip(g, w_items):
            assert abs(d.score - winfo[0]) <= 1e-3 * max(1.0, abs(winfo[0]))
        groups = _tie_groups(w_all)
        assert len({d.idx for d in g}) == len(g)
        for i, d in enumerate(g):
            assert d.idx in groups[i], (i, d.idx, sorted(groups[i]))
        assert g[0].docid == f"d{g[0].idx}"
        if not first_stage_only:
            assert g[0].raw_tokens() == [2] + orc.get_doc(g[0].idx)[:-1]   # `full` of keys.py:388, retrieval.py:685


class _WordTokenizer:
    """toy stand-in for the BART tokenizer: token id <-> "w<id>" words (enough for detokenisation)"""

    def decode(self, ids, skip_special_tokens=False, clean_up_tokenization_spaces=False):
        return " ".join(f"w{int(t)}" for t in ids if not (skip_special_tokens and int(t) in (0, 1, 2)))


@pytest.mark.parametrize("jobs", [1, 2])
def test_search_detokenizes_title_and_body(jobs, monkeypatch):
    """``search()`` always detokenises (reference retrieval.py:644-647,693-712): the retrieved passage is split at
    the title delimiter -- every 'title @@ body' passage holds one -- through the tokens the full scoring
    extracted (``doc._raw_tokens``), inline (jobs=1) and through worker processes."""
    from seal_amd import FMIndex
    from seal_amd import retrieval
    from seal_amd.retrieval import SEALSearcher
    from tests.helpers import make_docs, tiny_bart
    vocab = 120
    dev = torch.device("cuda:0")
    docs = make_docs(5, 200, vocab - 8, min_len=6, max_len=18, title_sep=TITLE_EOS)
    ix = FMIndex()
    ix.initialize(docs)
    monkeypatch.setattr(retrieval, "TITLE_MAX_LENGTH", 8)
    s = SEALSearcher(ix, _WordTokenizer(), tiny_bart(vocab).to(dev), backbone="bart-tiny", length=6, beam=4, batch_size=2,
                     add_query_to_keys=False, jobs=jobs, title_eos_token_id=TITLE_EOS, code_eos_token_id=vocab - 6,
                     code_bos_token_id=TITLE_EOS,
                     marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5]})
    rng = np.random.default_rng(1)
    query = [0] + rng.integers(4, vocab - 8, size=6).tolist() + [2]
    found = s.search(query, k=5)
    assert len(found) > 0
    for d in found:
        toks = [2] + docs[d.idx][:-1]                    # `full` of keys.py:388
        i = toks.index(TITLE_EOS)
        title, body = d.text()
        assert title == " ".join(f"w{t}" for t in toks[:i] if t > 2)
        assert body == " ".join(f"w{t}" for t in toks[i + 1:] if t > 2)


@pytest.mark.parametrize("geom", MODEL_GEOMETRIES, ids=MODEL_IDS)
def test_overlapped_batches_give_the_results_of_sequential_batches(geom, monkeypatch):
    """``overlap`` (the default: the next batches' decodes enqueued ahead of this batch's rescoring / aggregation; the decode and
    the rescoring phase alternating on the GPU -- ``exclusive_gemm_streams`` -- or sharing it as in round 3; one or two batches
    of decodes ahead) returns exactly what one batch after the other returns: same documents, bit-equal scores, same keys, in order"""
    from seal_amd import FMIndex
    from seal_amd import retrieval
    from seal_amd.retrieval import SEALSearcher
    from tests.helpers import make_docs, tiny_bart
    vocab = 120
    dev = torch.device("cuda:0")
    docs = make_docs(5, 300, vocab - 8, min_len=6, max_len=18, title_sep=TITLE_EOS)
    ix = FMIndex()
    ix.initialize(docs)
    rng = np.random.default_rng(3)
    queries = [[0] + rng.integers(4, vocab - 8, size=int(rng.integers(4, 9))).tolist() + [2] for _ in range(11)]
    monkeypatch.setattr(retrieval, "TITLE_MAX_LENGTH", 8)
    model = tiny_bart(vocab, **geom).to(dev)
    out = {}
    for depth, overlap, exclusive in ((1, False, True), (2, True, True), (1, True, True), (3, True, True), (1, True, False), (2, True, False)):
        s = SEALSearcher(ix, None, model, backbone="bart-tiny", length=6, beam=4, batch_size=2, add_query_to_keys=False,
                         detokenize=False, overlap_depth=depth, overlap=overlap, exclusive_gemm_streams=exclusive, include_keys=True,
                         title_eos_token_id=TITLE_EOS, code_eos_token_id=vocab - 6, code_bos_token_id=TITLE_EOS,
                         marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5]})
        assert s._overlapped() is overlap
        if not exclusive:
            # two library-GEMM streams at once re-arm round 3's stall: refused without the explicit switch (tiny GEMMs here: safe to run)
            with pytest.raises(RuntimeError, match="i_know_two_gemm_streams_can_stall"):
                s.batch_search(queries, k=10)
            s.i_know_two_gemm_streams_can_stall = True
        for rep in range(2):           # the second call reuses the captured graphs and buffers
            res = s.batch_search(queries, k=10)
            out[(depth, overlap, exclusive, rep)] = [[(d.idx, d.score, list(d.raw_tokens()), d.keys) for d in docs_] for docs_ in res]
    base = out[(1, False, True, 0)]
    assert sum(len(r) for r in base) > 30
    for key, val in out.items():
        assert val == base, key


@pytest.mark.parametrize("variant", ["unmarked_rescoring", "separate_scorer", "forced_second_token"])
def test_overlapped_search_with_a_forward_after_the_rescoring_yield(variant, monkeypatch):
    """configurations whose model forwards run AFTER the batch's marked rescoring is enqueued (reference retrieval.py:265-281 un-marked
    rescoring; keys.py:145-176 unigram scores from a separate scorer model or behind a forced second token): the overlapped search
    fences those forwards like the rescoring itself (one library-GEMM stream at a time) and returns what the sequential search returns"""
    from seal_amd import FMIndex
    from seal_amd import retrieval
    from seal_amd.retrieval import SEALSearcher
    from tests.helpers import make_docs, tiny_bart
    vocab = 120
    dev = torch.device("cuda:0")
    docs = make_docs(5, 300, vocab - 8, min_len=6, max_len=18, title_sep=TITLE_EOS)
    ix = FMIndex()
    ix.initialize(docs)
    rng = np.random.default_rng(5)
    queries = [[0] + rng.integers(4, vocab - 8, size=int(rng.integers(4, 9))).tolist() + [2] for _ in range(7)]
    monkeypatch.setattr(retrieval, "TITLE_MAX_LENGTH", 8)
    model = tiny_bart(vocab).to(dev)
    extra, scorer = {}, None
    if variant == "unmarked_rescoring":
        extra = dict(use_markers=False)
    elif variant == "separate_scorer":
        scorer = tiny_bart(vocab, seed=11).to(dev)
    else:
        extra = dict(force_decoding_second_token=int(docs[0][0]))
    gates = []
    out = {}
    for overlap in (False, True):
        s = SEALSearcher(ix, None, model, bart_scorer_model=scorer, backbone="bart-tiny", length=6, beam=4, batch_size=2, add_query_to_keys=False,
                         detokenize=False, overlap=overlap, title_eos_token_id=TITLE_EOS, code_eos_token_id=vocab - 6, code_bos_token_id=TITLE_EOS,
                         marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5]}, **extra)
        if overlap:
            real = retrieval.rk.compute_unigram_scores if variant != "unmarked_rescoring" else retrieval.rk.rescore_keys
            name = "compute_unigram_scores" if variant != "unmarked_rescoring" else "rescore_keys"

            def spy(*a, _real=real, **kw):
                gates.append(s.__dict__.get("_gemm_gate") is not None)      # the gate is still installed when the late forward runs
                return _real(*a, **kw)
            monkeypatch.setattr(retrieval.rk, name, spy)
        res = s.batch_search(queries, k=10)
        out[overlap] = [[(d.idx, d.score) for d in docs_] for docs_ in res]
    assert gates and all(gates)
    assert out[True] == out[False] and sum(len(r) for r in out[True]) > 10


def _tiny_gpu_searcher(ix, model, vocab, **kw):
    from seal_amd.retrieval import SEALSearcher
    return SEALSearcher(ix, None, model, backbone="bart-tiny", length=6, beam=4, batch_size=2, detokenize=False,
                        title_eos_token_id=TITLE_EOS, code_eos_token_id=vocab - 6, code_bos_token_id=TITLE_EOS,
                        marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5]},
                        **{"add_query_to_keys": False, **kw})


def _short_titles(monkeypatch):
    # the title decode length (15) is hard-wired in the reference; shortened for the tiny corpora on both sides
    from seal_amd import retrieval
    monkeypatch.setattr(retrieval, "TITLE_MAX_LENGTH", 8)


def test_sharded_search_over_rccl_with_one_rank_equals_the_local_search(monkeypatch):
    """the N > 1 path as the driver's multi-GPU bench runs it -- ``sharded_batch_search``: shard, search, ONE
    ``all_gather_into_tensor`` of [q, k, 2] float64 on the device -- on backend "nccl" (= RCCL), world size 1: the
    collective, its device buffers and the process-group set-up are the real ones; the result must be bit-equal to the
    plain local search"""
    import os
    import torch.distributed as dist
    from seal_amd import FMIndex
    from seal_amd.distributed import pack_topk, sharded_batch_search
    from tests.helpers import make_docs, tiny_bart
    from tests.test_distributed_gloo import _free_port
    vocab = 120
    dev = torch.device("cuda:0")
    docs = make_docs(5, 200, vocab - 8, min_len=6, max_len=18, title_sep=TITLE_EOS)
    ix = FMIndex()
    ix.initialize(docs)
    _short_titles(monkeypatch)
    s = _tiny_gpu_searcher(ix, tiny_bart(vocab, d_model=128, heads=2).to(dev), vocab)
    rng = np.random.default_rng(5)
    queries = [[0] + rng.integers(4, vocab - 8, size=int(rng.integers(4, 9))).tolist() + [2] for _ in range(5)]
    want = pack_topk(s.batch_search(queries, k=10), 10)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        got = sharded_batch_search(s, queries, k=10, device=dev)
        assert got.is_cuda and got.shape == (5, 10, 2)
        assert torch.equal(got.cpu(), want)
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                       # the bench's max-over-ranks timing collective
        assert float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()


def test_query_longer_than_the_fused_kernels_reach_falls_back_loudly_and_ranks_the_same(monkeypatch, caplog):
    """a query of more than 64 encoder tokens is past the fused step kernels (one lane per encoder position): the whole
    batch then decodes through the torch-op path -- with ONE warning, not silently -- and ranks like the scalar pipeline"""
    import logging
    from oracle.seal_oracle import OracleFMIndex
    from seal_amd import FMIndex
    from tests.helpers import make_docs, tiny_bart
    vocab = 120
    dev = torch.device("cuda:0")
    docs = make_docs(5, 200, vocab - 8, min_len=6, max_len=18, title_sep=TITLE_EOS)
    ix, orc = FMIndex(), OracleFMIndex()
    ix.initialize(docs)
    orc.initialize(docs)
    _short_titles(monkeypatch)
    geom = dict(d_model=128, heads=2, max_positions=128)
    s = _tiny_gpu_searcher(ix, tiny_bart(vocab, **geom).to(dev), vocab)
    rng = np.random.default_rng(9)
    queries = [[0] + rng.integers(4, vocab - 8, size=n).tolist() + [2] for n in (70, 5)]      # 76 and 11 encoder tokens with the markers
    with caplog.at_level(logging.WARNING, logger="seal_amd.bart_decoder"):
        got = s.batch_search(queries, k=10)
        got2 = s.batch_search(queries, k=10)
    assert s.bart_model._seal_step_decoder._st.fused is False
    assert sum("torch-op path" in r.getMessage() for r in caplog.records) == 1
    assert [[(d.idx, d.score) for d in g] for g in got] == [[(d.idx, d.score) for d in g] for g in got2]
    want = _oracle_pipeline(tiny_bart(vocab, **geom), orc, queries, 4, 6, vocab, False)
    for g, w in zip(got, want):
        w_all = list(w.items())
        groups = _tie_groups(w_all)
        assert len(g) == min(10, len(w_all)) > 0
        for i, d in enumerate(g):
            assert d.idx in groups[i] and abs(d.score - w_all[i][1][0]) <= 1e-3 * max(1.0, abs(w_all[i][1][0]))


def test_searcher_document_ids_on_a_corpus_where_ties_are_rare(monkeypatch):
    """end-to-end document ids against the scalar pipeline at a size where equal scores are the exception: 1 500
    documents over a 1 000-token vocabulary, the tie tolerance tightened to 1e-5 relative (fp32 model scores from two
    devices) -- almost every rank position then admits exactly one document id"""
    from oracle.seal_oracle import OracleFMIndex
    from seal_amd import FMIndex
    from tests.helpers import make_docs, tiny_bart
    vocab = 1000
    dev = torch.device("cuda:0")
    docs = make_docs(21, 1500, vocab - 8, min_len=8, max_len=30, title_sep=TITLE_EOS)
    ix, orc = FMIndex(), OracleFMIndex()
    ix.initialize(docs)
    orc.initialize(docs)
    _short_titles(monkeypatch)
    geom = dict(d_model=128, heads=2)
    s = _tiny_gpu_searcher(ix, tiny_bart(vocab, **geom).to(dev), vocab, add_query_to_keys=True)
    rng = np.random.default_rng(2)
    queries = []
    for _ in range(4):                           # queries made of corpus n-grams, so that keys are found
        d = docs[int(rng.integers(len(docs)))]
        a = int(rng.integers(0, max(1, len(d) - 6)))
        queries.append([0] + d[a:a + 5] + rng.integers(4, vocab - 8, size=2).tolist() + [2])
    got = s.batch_search(queries, k=20)
    assert s.bart_model._seal_step_decoder._st.fused is True
    want = _oracle_pipeline(tiny_bart(vocab, **geom), orc, queries, 4, 6, vocab, False, query_keys=True)
    single = total = 0
    for g, w in zip(got, want):
        w_all = list(w.items())
        groups = _tie_groups(w_all, rel=1e-5)
        assert len(g) == min(20, len(w_all)) > 0
        for i, d in enumerate(g):
            assert d.idx in groups[i], (i, d.idx, sorted(groups[i]))
            assert abs(d.score - w_all[i][1][0]) <= 1e-4 * max(1.0, abs(w_all[i][1][0]))
            single += len(groups[i]) == 1
            total += 1
    assert single >= 0.9 * total, (single, total)


@pytest.mark.parametrize("decode_code,stop_at_count", [(False, 0), (True, 0), (False, 4), (True, 3)],
                         ids=["body+title", "body+title+code", "body+title-stop_at_count", "body+title+code-stop_at_count"])
def test_joint_decode_of_a_batch_returns_what_separate_decodes_return(decode_code, stop_at_count, monkeypatch):
    """``joint_decode`` (default): the body / title (/ code) decodes of a batch as ONE loop of stacked rows; off: one
    ``fm_index_generate`` after the other as the reference does.  Same documents (ties at 1e-5 relative aside), scores
    within 1e-4 relative: the GEMMs run at another height, nothing else differs.  With ``stop_at_count`` > 0 the value is the
    body decode's alone on both paths (reference retrieval.py:70-83; titles / codes run with 0)."""
    from seal_amd import FMIndex
    from tests.helpers import make_docs, tiny_bart
    vocab = 120
    dev = torch.device("cuda:0")
    docs = make_docs(5, 300, vocab - 8, min_len=6, max_len=18, title_sep=TITLE_EOS)
    ix = FMIndex()
    ix.initialize(docs)
    _short_titles(monkeypatch)
    model = tiny_bart(vocab, d_model=128, heads=2).to(dev)
    rng = np.random.default_rng(8)
    queries = [[0] + rng.integers(4, vocab - 8, size=int(rng.integers(4, 9))).tolist() + [2] for _ in range(5)]
    out = {}
    for joint in (True, False):
        s = _tiny_gpu_searcher(ix, model, vocab, joint_decode=joint, add_query_to_keys=True, decode_code=decode_code, partial_code=decode_code,
                               overlap=False, stop_at_count=stop_at_count)
        s.marker_token_ids["code"] = [vocab - 2, vocab - 7]
        calls = []
        from seal_amd import beam_search
        real = beam_search.constrained_beam_search_groups
        monkeypatch.setattr(beam_search, "constrained_beam_search_groups", lambda dec, specs, *a, **kw: (calls.append(len(specs)), real(dec, specs, *a, **kw))[1])
        res = s.batch_search(queries, k=10)
        monkeypatch.setattr(beam_search, "constrained_beam_search_groups", real)
        n_dec = 3 if decode_code else 2
        assert calls == ([n_dec] * 3 if joint else [1] * (3 * n_dec))           # 3 batches of <= 2 queries
        out[joint] = [[(d.idx, d.score) for d in docs_] for docs_ in res]
    for a, b in zip(out[True], out[False]):
        assert len(a) == len(b) > 0
        for i, ((da, sa), (db, sb)) in enumerate(zip(a, b)):
            assert abs(sa - sb) <= 1e-4 * max(1.0, abs(sb))
            if da != db:                                   # a swap inside a tie group only
                assert any(abs(sb - s2) <= 1e-5 * max(1.0, abs(sb)) and d2 == da for d2, s2 in b)
