"""Vectors produced by the reference's OWN Python (tests/golden/ref_*.json, written by
tests/golden/make_reference_golden.py from /root/reference/seal/{index,keys,beam_search}.py running on the oracle's
model of the C++ layer) against the oracle restatements and the product's host logic.  Everything above the
SWIG boundary is pinned to the reference's real code here; the HIP index is held to the index / mask vectors
directly (-m gpu) and joins the rest through the -m gpu tests that compare it with the same oracle."""
import json
import os

import numpy as np
import pytest

from oracle.beam_oracle import oracle_logits_mask
from oracle.keys_oracle import oracle_aggregate_evidence, oracle_deduplicate, oracle_strip
from oracle.seal_oracle import OracleFMIndex
from seal_amd.keys import aggregate_evidence, deduplicate, strip
from tests.helpers import OracleBatchIndex

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def _unhex(x):
    return float.fromhex(x)


def _as_golden(results):
    out = []
    for doc, info in results.items():
        out.append({"doc": int(doc), "score": float(info[0]).hex(),
                    "keys": [[[int(t) for t in k], float(s).hex()] for k, s in info[1]],
                    "tokens": [int(t) for t in info[3]],
                    "best": [[int(t) for t in info[4][0]], float(info[4][1]).hex()]})
    return out


AGG = _load("ref_aggregate_evidence.json")["cases"]


@pytest.mark.parametrize("case_no", range(len(AGG)))
def test_aggregate_evidence_equals_the_reference(case_no):
    case = AGG[case_no]
    orc = OracleFMIndex()
    orc.initialize(case["docs"])
    keys = [(list(k), _unhex(s)) for k, s in case["keys"]]
    us = None if case["unigram_scores"] is None else [_unhex(x) for x in case["unigram_scores"]]
    want_ngrams = [[list(k), s] for k, s in case["all_ngrams"]]
    for name, run in (("oracle", lambda: oracle_aggregate_evidence(keys, unigram_scores=us, index=orc, **case["kwargs"])),
                      ("product host logic", lambda: aggregate_evidence(keys, unigram_scores=us, index=OracleBatchIndex(orc), **case["kwargs"]))):
        results, all_ngrams = run()
        assert [[list(k), float(s).hex()] for k, s in all_ngrams.items()] == want_ngrams, name
        got = _as_golden(results)
        assert [g["doc"] for g in got] == [w["doc"] for w in case["results"]], name       # ranking
        assert got == case["results"], name                                                 # scores (bit-exact), keys, tokens, best key


def test_strip_and_deduplicate_equal_the_reference():
    h = _load("ref_helpers.json")
    for c in h["strip"]:
        assert oracle_strip(list(c["seq"]), c["start"], c["end"]) == c["out"]
        assert strip(list(c["seq"]), c["start"], c["end"]) == c["out"]
    for c in h["deduplicate"]:
        typed = [(x[0], x[1]) if isinstance(x[0], float) else x for x in c["items"]]
        want = [typed[i] for i in c["kept_positions"]]
        assert oracle_deduplicate(typed) == want
        assert deduplicate(typed) == want


def test_query_decomposition_equals_the_reference(monkeypatch):
    """add_query_to_keys: every word 1..n-gram of the query in every capitalisation pattern (keys.py:38-51), with
    whitespace words standing in for spaCy's tokenizer on both sides"""
    from seal_amd import query_keys

    class Word:
        def __init__(self, text):
            self.text = text
    monkeypatch.setattr(query_keys, "_word_tokenizer", lambda q: [Word(w) for w in q.split()])
    for c in _load("ref_helpers.json")["decompose"]:
        assert sorted(query_keys.decompose_query_into_keys(c["query"], c["length"])) == c["keys_sorted"], c["query"]


IDX = _load("ref_index_and_mask.json")["cases"]


def _check_index_and_masks(ix, case, allowed_tokens):
    """seal/index.py's wrapper logic and IndexBasedLogitsProcessor.__call__ as the reference computed them"""
    assert ix.size() == case["size"] and len(ix) == case["len"] and ix.n_docs == case["n_docs"]
    assert list(ix.beginnings) == case["beginnings"]
    assert sorted(ix.occurring) == case["occurring_sorted"]
    assert list(ix.occurring_distinct) == case["occurring_distinct"] and list(ix.occurring_counts) == case["occurring_counts"]
    for seq, rng, cnt in case["ranges"]:
        lo, hi = ix.get_range(list(seq))
        assert [int(lo), int(hi)] == rng and int(ix.get_count(list(seq))) == cnt
    for seq, conts in case["continuations"]:
        assert sorted(int(t) for t in ix.get_continuations(list(seq))) == conts
    for row, tok, doc in case["rows"]:
        assert int(ix.get_token_index_from_row(row)) == tok and int(ix.get_doc_index_from_row(row)) == doc
    assert [[int(t) for t in ix.get_doc(i)] for i in range(ix.n_docs)] == case["docs_back"]
    for lo, hi, want in case["distinct_count"]:
        toks, cnts = ix.get_distinct_count(lo, hi)
        assert [[int(t) for t in toks], [int(c) for c in cnts]] == want
    assert case["distinct_count_multi_equals_single"]
    for m in case["masks"]:
        kw = dict(m["kwargs"])
        eos = kw.pop("eos_token_id", 2)
        assert allowed_tokens(m["input_ids"], eos, kw) == m["allowed"], (m["kwargs"], m["input_ids"])


@pytest.mark.parametrize("case_no", range(len(IDX)))
def test_oracle_index_and_mask_equal_the_reference_python(case_no):
    case = IDX[case_no]
    orc = OracleFMIndex()
    orc.initialize(case["docs"])

    def allowed_tokens(input_ids, eos, kw):
        got = oracle_logits_mask(orc, input_ids, case["vocab"], 4, pad_token_id=1, eos_token_id=eos, **kw)
        return [np.flatnonzero(r).tolist() for r in got]
    _check_index_and_masks(orc, case, allowed_tokens)


@pytest.mark.gpu
@pytest.mark.parametrize("case_no", range(len(IDX)))
def test_hip_index_and_mask_equal_the_reference_python(case_no):
    """the HIP index behind the reference's FMIndex API, and the product's IndexBasedLogitsProcessor, against the same
    reference-produced vectors"""
    import torch
    from seal_amd import FMIndex
    from seal_amd.beam_search import IndexBasedLogitsProcessor
    case = IDX[case_no]
    ix = FMIndex()
    ix.initialize(case["docs"])
    dev = torch.device("cuda:0")

    def allowed_tokens(input_ids, eos, kw):
        proc = IndexBasedLogitsProcessor(ix, 4, pad_token_id=1, eos_token_id=eos, **kw)
        out = proc(torch.tensor(input_ids, device=dev), torch.zeros(len(input_ids), case["vocab"], device=dev))
        return [torch.nonzero(r == 0.0).flatten().tolist() for r in out]
    _check_index_and_masks(ix, case, allowed_tokens)


@pytest.mark.gpu
@pytest.mark.parametrize("case_no", range(len(IDX)))
def test_reference_call_pattern_on_the_hip_swig_shim(case_no):
    """``seal/index.py`` on top of ``seal_amd.cpp_modules.fm_index`` (the class that takes
    the place of the SWIG module, reference index.py:13-14).  The reference's sources cannot travel to the GPU box
    and this container has no GPU, so the layer above the SWIG surface is its restatement (oracle/seal_oracle.py
    ``OracleFMIndex``: the SCALAR call pattern of index.py:39-204 -- one ``backward_search_step`` per token,
    ``distinct_count_multi`` + python unzip, ``locate`` + bisect, ``extract_text``) re-based onto the HIP shim's ten
    SWIG methods, and the reference's mask logic (beam_search.py:62-140, restated in oracle/beam_oracle.py) on top of
    that -- held to the vectors the reference's own classes produced."""
    from seal_amd.cpp_modules.fm_index import FMIndex as HipSwigFMIndex
    case = IDX[case_no]
    layer = {k: v for k, v in OracleFMIndex.__dict__.items() if k not in ("__dict__", "__weakref__", "__init__", "initialize", "__doc__")}

    def init(self):
        HipSwigFMIndex.__init__(self)
        self.beginnings, self.occurring, self.occurring_distinct, self.occurring_counts, self.labels = [0], set(), [], [], None

    def initialize(self, sequences, in_memory=False):          # reference index.py:39-66
        data, occurring = [], set()
        for seq in sequences:
            self.beginnings.append(self.beginnings[-1] + len(seq))
            occurring |= set(seq)
            data.extend(x + 10 for x in reversed(seq))
        self.occurring = list(occurring)
        HipSwigFMIndex.initialize(self, data)
        self.occurring_distinct, self.occurring_counts = self.get_distinct_count(0, len(self))
    cls = type("ReferenceLayerOnHipShim", (HipSwigFMIndex,), {**layer, "__init__": init, "initialize": initialize})
    ix = cls()
    ix.initialize(case["docs"])
    assert type(ix).get_range is OracleFMIndex.get_range and type(ix).backward_search_step is HipSwigFMIndex.backward_search_step

    def allowed_tokens(input_ids, eos, kw):
        got = oracle_logits_mask(ix, input_ids, case["vocab"], 4, pad_token_id=1, eos_token_id=eos, **kw)
        return [np.flatnonzero(r).tolist() for r in got]
    _check_index_and_masks(ix, case, allowed_tokens)


def test_model_side_scores_equal_the_reference():
    """compute_unigram_scores and rescore_keys (prefix-sharing and row-per-key) against the numbers the
    reference's own functions produced with the same seeded tiny BART on CPU (fp32: 1e-5, the summation
    order differs)"""
    import torch
    from seal_amd.keys import compute_unigram_scores, rescore_keys
    from tests.helpers import tiny_bart
    g = _load("ref_model_side.json")
    model = tiny_bart(vocab=g["vocab"], seed=g["model_seed"])
    inputs = g["inputs"]
    with torch.no_grad():
        got = compute_unigram_scores(model, inputs, None, tolist=True)
        want = [[_unhex(x) for x in row] for row in g["unigram_scores"]]
        assert np.allclose(np.asarray(got, dtype=np.float64), np.asarray(want), atol=1e-5, rtol=0, equal_nan=True)
        got = compute_unigram_scores(model, inputs, None, tolist=True, prefix=[5, 9])
        want = [[_unhex(x) for x in row] for row in g["unigram_scores_prefix"]]
        assert np.allclose(np.asarray(got, dtype=np.float64), np.asarray(want), atol=1e-5, rtol=0, equal_nan=True)
        decoded = [[(_unhex(x[0]), x[1]) if isinstance(x[0], str) else x for x in kk] for kk in g["decoded"]]
        for run in g["rescore"]:
            for share in (True, False):
                res = rescore_keys(model, inputs, decoded, batch_size=4, share_prefixes=share, **run["kwargs"])
                assert len(res) == len(run["scores"])
                for per_query, want_q in zip(res, run["scores"]):
                    assert [list(k) for _, k in per_query] == [k for _, k in want_q], run["kwargs"]     # same keys, same order
                    a = np.asarray([s for s, _ in per_query], dtype=np.float64)
                    b = np.asarray([_unhex(s) for s, _ in want_q])
                    assert np.allclose(a, b, atol=1e-5, rtol=0), (run["kwargs"], share)


BEAM = _load("ref_beam_search.json")
BEAM_DH64 = _load("ref_beam_search_dh64.json")       # the same generator at head_dim 64 (the fused decoder kernels' geometry; -m gpu twins)
BEAM_CASES = [(BEAM, i) for i in range(len(BEAM["cases"]))] + [(BEAM_DH64, i) for i in range(len(BEAM_DH64["cases"]))]
BEAM_IDS = ["dh8-%d" % i for i in range(len(BEAM["cases"]))] + ["dh64-%d" % i for i in range(len(BEAM_DH64["cases"]))]


@pytest.mark.parametrize("BEAM,case_no", BEAM_CASES, ids=BEAM_IDS)
def test_decode_equals_the_reference_loop(BEAM, case_no):
    """hypotheses of the reference's whole fm_index_generate(keep_history=True) -- its beam loop, scorer with
    memory and constraint processor, run for real by the generator -- against the oracle's restatement of that
    loop and against the product's tensorised loop (on CPU through the oracle-backed constraint; the HIP
    constraint joins through the -m gpu tests).  Compared the way the searcher consumes them (strip, count > 0,
    first occurrence): that is the level at which the reference itself is deterministic, because torch.topk
    orders the -inf candidates of a short beam arbitrarily.  Scores: 1e-4 (north_star)."""
    import torch
    from oracle.beam_oracle import oracle_fm_index_generate
    from seal_amd.beam_search import fm_index_generate
    from tests.helpers import OracleLogitsProcessor, hf_logits_fn, tiny_bart, valid_set
    vocab = BEAM["vocab"]
    case = BEAM["cases"][case_no]
    kw = dict(case["kwargs"])
    orc = OracleFMIndex()
    orc.initialize(BEAM["docs"])
    bart = tiny_bart(vocab, **BEAM.get("model_kw", {}))
    enc_ids = torch.tensor(BEAM["enc_ids"])
    enc_mask = torch.ones_like(enc_ids)
    K, eos = kw["num_beams"], kw.get("eos_token_id", 2)
    want = [[(_unhex(s), toks) for s, toks in per_query] for per_query in case["hypotheses"]]
    opts = dict(force_decoding_from=kw.get("force_decoding_from"), stop_at_count=kw.get("stop_at_count", 0),
                always_allow_eos=kw.get("always_allow_eos", False))
    by_oracle = oracle_fm_index_generate(hf_logits_fn(bart, enc_ids, enc_mask, K), orc, len(enc_ids), K, kw["max_length"], vocab,
                                         decoder_start_token_id=2, pad_token_id=1, eos_token_id=eos, length_penalty=kw["length_penalty"],
                                         disable_fm_index=kw.get("disable_fm_index", False), **opts)
    proc = OracleLogitsProcessor(orc, K, vocab, pad_token_id=1, eos_token_id=eos, **opts)
    by_product = fm_index_generate(bart, None, enc_ids, enc_mask, min_length=1, keep_history=True, constrained_decoding_processor=proc, **kw)
    title_eos = eos if kw.get("force_decoding_from") else None
    for name, got in (("oracle", by_oracle), ("product loop", by_product)):
        assert len(got) == len(want)
        for g, w in zip(got, want):
            gv, wv = valid_set(g, orc, title_eos), valid_set(w, orc, title_eos)
            assert set(gv) == set(wv), name
            for k in wv:
                assert abs(gv[k][0] - wv[k][0]) <= 1e-4, (name, k)
            if kw.get("disable_fm_index"):                  # no -inf candidates without the constraint: the full list, in order
                assert [list(t) for _, t in g] == [list(t) for _, t in w], name
                assert np.allclose([s for s, _ in g], [s for s, _ in w], atol=1e-4, rtol=0), name


SEARCH = _load("ref_searcher.json")


@pytest.mark.parametrize("run_no", range(len(SEARCH["runs"])))
def test_search_pipeline_equals_the_reference_searcher(run_no):
    """the reference's SEALSearcher.batch_search run end to end by the generator (process_batch: body decode,
    post-filters, rescoring, title decode, title filters, rescoring, dedup, unigram scores; then aggregate_evidence
    with the searcher's parameters) against the scalar pipeline that tests/test_gpu_search.py holds the HIP
    searcher to.  Keys: same set, scores 1e-4; ranked documents: same ids wherever the reference's scores are
    separated by more than the tolerance, scores 1e-4 relative (fp32 model scores enter a float64 pipeline)."""
    from tests.helpers import tiny_bart
    from tests.test_gpu_search import _oracle_pipeline
    run = SEARCH["runs"][run_no]
    vocab, K, length = SEARCH["vocab"], SEARCH["beam"], SEARCH["length"]
    orc = OracleFMIndex()
    orc.initialize(SEARCH["docs"])
    results, keys = _oracle_pipeline(tiny_bart(vocab), orc, SEARCH["queries"], K, length, vocab, False,
                                     title_length=run["title_length"], return_keys=True, query_keys=run["add_query_to_keys"],
                                     decode_code=run.get("decode_code", False))
    for got, got_keys, want in zip(results, keys, run["queries"]):
        want_keys = {tuple(n): _unhex(s) for n, s in want["keys"]}
        have_keys = {tuple(n): s for n, s in got_keys}
        assert set(have_keys) == set(want_keys)
        for n in want_keys:
            assert abs(have_keys[n] - want_keys[n]) <= 1e-4, n
        ranked = list(got.items())[:10]
        assert len(ranked) == len(want["ranked"]) > 0
        w_scores = [_unhex(d["score"]) for d in want["ranked"]]
        for (doc, info), w in zip(ranked, w_scores):
            assert abs(info[0] - w) <= 1e-4 * max(1.0, abs(w))
        if all(abs(a - b) > 1e-4 * max(1.0, abs(a)) for a, b in zip(w_scores, w_scores[1:])):
            assert [doc for doc, _ in ranked] == [d["doc"] for d in want["ranked"]]
        for d in want["ranked"]:
            assert d["docid"] == f"d{d['doc']}"
            assert d["raw_tokens"] == [2] + orc.get_doc(d["doc"])[:-1]        # keys.py:388 / retrieval.py:685
            assert d["raw_tokens"] == got[d["doc"]][3] if d["doc"] in got else True


def test_fairseq_checkpoint_loader_equals_the_reference(tmp_path):
    """seal_amd.utils.load_state_dict_from_fairseq_checkpoint leaves the model in the state the reference's loader
    (seal/utils.py:42-50) leaves it in: every tensor (shape, sum, sum of magnitudes), the lm_head tied to the
    embedding with the appended zero row, and the logits of a fixed input"""
    import torch
    from seal_amd.utils import load_state_dict_from_fairseq_checkpoint
    from tests.helpers import synthetic_fairseq_checkpoint, tiny_bart
    g = _load("ref_checkpoint.json")
    model = tiny_bart(g["vocab"], seed=g["target_seed"])
    path = str(tmp_path / "ckpt.pt")
    synthetic_fairseq_checkpoint(path, vocab=g["vocab"], seed=g["checkpoint_seed"])
    load_state_dict_from_fairseq_checkpoint(model, path)
    model.eval()
    state = model.state_dict()
    assert set(state) == set(g["state"])
    for k, (shape, total, magnitude) in g["state"].items():
        v = state[k]
        assert list(v.shape) == shape, k
        assert v.double().sum().item() == _unhex(total) and v.double().abs().sum().item() == _unhex(magnitude), k
    assert torch.equal(model.lm_head.weight, model.model.shared.weight) == g["lm_head_is_embedding"]
    enc = torch.tensor([[0, 5, 17, 33, 2], [0, 9, 9, 2, 1]])
    with torch.no_grad():
        logits = model(input_ids=enc, attention_mask=(enc != 1).long(), decoder_input_ids=torch.tensor([[2, 5], [2, 9]])).logits
    want = np.asarray([_unhex(x) for x in g["logits_sample"]])
    assert np.allclose(logits[:, -1, :16].flatten().double().numpy(), want, atol=1e-6, rtol=0, equal_nan=True)


@pytest.mark.parametrize("title_length,jobs,query_keys", [(8, 1, None), (15, 1, None), (8, 2, None), (8, 1, "strings"), (8, 1, "token ids"), (8, 1, "code")])
def test_product_searcher_equals_the_reference_searcher(title_length, jobs, query_keys, monkeypatch):
    """the PRODUCT's SEALSearcher.batch_search, all of its Python (key generation recipe, batched post-filters,
    prefix-sharing rescoring, batched evidence aggregation with the native host routines, worker processes,
    SEALDocument) run on CPU -- index queries answered by the oracle, the constraint by the oracle's mask -- against
    what the reference's own SEALSearcher returned for the same corpus, model and queries"""
    from seal_amd import retrieval
    from seal_amd.retrieval import SEALSearcher
    from tests.helpers import OracleLogitsProcessor, tiny_bart
    decode_code = query_keys == "code"        # the run with the code decode switched on (partial_code: the corpus has no code sections)
    query_keys = None if decode_code else query_keys
    run = [r for r in SEARCH["runs"] if r["title_length"] == title_length and r["add_query_to_keys"] == (query_keys is not None)
           and bool(r.get("decode_code")) == decode_code][0]
    vocab, K, length, title_eos = SEARCH["vocab"], SEARCH["beam"], SEARCH["length"], SEARCH["title_eos"]
    orc = OracleFMIndex()
    orc.initialize(SEARCH["docs"])

    class CpuIndex(OracleBatchIndex):
        labels = None

        @property
        def n_docs(self):
            return self.orc.n_docs

        def get_doc(self, i):
            return self.orc.get_doc(i)

        def get_range(self, seq):
            return self.orc.get_range(list(seq))
    index = CpuIndex(orc)
    index.labels = [f"d{i}" for i in range(len(SEARCH["docs"]))]
    real = retrieval.fm_index_generate

    def generate(model, _index, *a, **kw):
        proc = OracleLogitsProcessor(orc, kw["num_beams"], vocab, pad_token_id=1, eos_token_id=kw.get("eos_token_id") or 2,
                                     force_decoding_from=kw.get("force_decoding_from"), stop_at_count=kw.get("stop_at_count", 0),
                                     always_allow_eos=kw.get("always_allow_eos", False))
        if kw.get("force_decoding_from"):        # the title decode length is a constant (15) in both code bases
            kw = {**kw, "max_length": title_length}
        return real(model, None, *a, constrained_decoding_processor=proc, **kw)
    monkeypatch.setattr(retrieval, "fm_index_generate", generate)
    # add_query_to_keys (the reference's default, retrieval.py:115-149): as the reference ran it -- query STRINGS, a
    # whitespace word tokenizer in spaCy's place, the toy BART tokenizer -- and through the product's token-id form
    # (seal_amd.query_keys.token_ngram_keys: with one token per word and case-insensitive tokens the same key set)
    tokenizer, queries = None, SEARCH["queries"]
    if query_keys == "strings":
        from seal_amd import query_keys as qk
        from tests.helpers import ToyTokenizer, split_words
        monkeypatch.setattr(qk, "_word_tokenizer", split_words)
        tokenizer, queries = ToyTokenizer(vocab), [" ".join(f"w{t}" for t in q[1:-1]) for q in SEARCH["queries"]]
    s = SEALSearcher(index, tokenizer, tiny_bart(vocab), backbone="bart-tiny", length=length, beam=K, batch_size=2,
                     add_query_to_keys=query_keys is not None, detokenize=False, jobs=jobs, title_eos_token_id=title_eos,
                     code_eos_token_id=vocab - 6, code_bos_token_id=title_eos, decode_code=decode_code, partial_code=decode_code,
                     marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5],
                                       "code": [vocab - 2, vocab - 7]})
    if query_keys is not None:       # the keys themselves: same n-grams, scores within fp32 noise (their order follows a python set's)
        for got_q, want_q in zip(s.batch_generate_keys(queries), run["queries"]):
            gk = {tuple(k): v for k, v in got_q[0]}
            wk = {tuple(k): _unhex(v) for k, v in want_q["keys"]}
            assert set(gk) == set(wk)
            assert all(abs(gk[k] - wk[k]) <= 1e-4 * max(1.0, abs(wk[k])) for k in wk)
    got = s.batch_search(queries, k=10)
    assert len(got) == len(run["queries"])
    for docs, want in zip(got, run["queries"]):
        w_scores = [_unhex(d["score"]) for d in want["ranked"]]
        assert len(docs) == len(w_scores) > 0
        for d, w in zip(docs, w_scores):
            assert abs(d.score - w) <= 1e-4 * max(1.0, abs(w))
        if all(abs(a - b) > 1e-4 * max(1.0, abs(a)) for a, b in zip(w_scores, w_scores[1:])):
            assert [d.idx for d in docs] == [d["doc"] for d in want["ranked"]]
            for d, w in zip(docs, want["ranked"]):
                assert d.docid == w["docid"]
                assert list(d.raw_tokens()) == w["raw_tokens"]


@pytest.mark.parametrize("jobs", [1, 2])
def test_product_search_detokenizes_through_the_title_delimiter(jobs, monkeypatch):
    """``SEALSearcher.search`` always detokenises (reference retrieval.py:644-647): ``split_tokens`` must work on the
    document tokens the full scoring hands over (``doc._raw_tokens``), inline and through worker processes --
    every 'title @@ body' passage holds the delimiter"""
    from seal_amd import retrieval
    from seal_amd.retrieval import SEALSearcher
    from tests.helpers import OracleLogitsProcessor, tiny_bart
    vocab, K, length, title_eos = SEARCH["vocab"], SEARCH["beam"], SEARCH["length"], SEARCH["title_eos"]
    orc = OracleFMIndex()
    orc.initialize(SEARCH["docs"])

    class CpuIndex(OracleBatchIndex):
        labels = None
        n_docs = property(lambda self: self.orc.n_docs)

        def get_doc(self, i):
            return self.orc.get_doc(i)

    class WordTokenizer:
        def decode(self, ids, skip_special_tokens=False, clean_up_tokenization_spaces=False):
            return " ".join(f"w{int(t)}" for t in ids if not (skip_special_tokens and int(t) in (0, 1, 2)))
    real = retrieval.fm_index_generate

    def generate(model, _index, *a, **kw):
        proc = OracleLogitsProcessor(orc, kw["num_beams"], vocab, pad_token_id=1, eos_token_id=kw.get("eos_token_id") or 2,
                                     force_decoding_from=kw.get("force_decoding_from"))
        if kw.get("force_decoding_from"):
            kw = {**kw, "max_length": 8}
        return real(model, None, *a, constrained_decoding_processor=proc, **kw)
    monkeypatch.setattr(retrieval, "fm_index_generate", generate)
    s = SEALSearcher(CpuIndex(orc), WordTokenizer(), tiny_bart(vocab), backbone="bart-tiny", length=length, beam=K, batch_size=2,
                     add_query_to_keys=False, jobs=jobs, title_eos_token_id=title_eos, code_eos_token_id=vocab - 6,
                     code_bos_token_id=title_eos,
                     marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5]})
    found = s.search(SEARCH["queries"][0], k=5)
    assert len(found) > 0
    for d in found:
        toks = [2] + orc.get_doc(d.idx)[:-1]
        title, body = d.text()
        if title_eos in toks:
            i = toks.index(title_eos)
            assert title == " ".join(f"w{t}" for t in toks[:i] if t > 2)
            assert body == " ".join(f"w{t}" for t in toks[i + 1:] if t > 2)
        else:
            assert title == "" and body == " ".join(f"w{t}" for t in toks if t > 2)
