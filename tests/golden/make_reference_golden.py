#!/usr/bin/env python
"""Generates tests/golden/ref_*.json by RUNNING THE REFERENCE'S OWN PYTHON (facebookresearch/SEAL,
/root/reference/seal/{index,keys,beam_search}.py) in this container.

What can and cannot be run here:
  * seal/cpp_modules (SWIG + sdsl-lite) cannot be built offline.  The reference's Python only reaches
    it through ten methods of `_FMIndex`; this script puts a stand-in module at
    `seal.cpp_modules.fm_index` whose `FMIndex` is the oracle's restatement of that C++ class
    (oracle/seal_oracle.py::CppFMIndex over oracle/fm_oracle.c).  So the vectors pin everything ABOVE
    the C++ boundary -- `seal.index.FMIndex` (shift / reversal / doc binning / distinct filters),
    `seal.keys.aggregate_evidence`, `strip`, `deduplicate`, and
    `seal.beam_search.IndexBasedLogitsProcessor.__call__` -- against the reference's real code, on top
    of the oracle's model of the C++ layer (which stays pinned by brute force only, tests/golden/fm_golden.json).
  * `more_itertools` is absent (only `chunked` / `ichunked` are imported: four-line stand-ins below), and
    the installed transformers (5.x) no longer has the generation internals beam_search.py imports
    at module level (BeamScorer & co, transformers.generation_utils): they are given placeholder
    names so that the module imports; `fm_index_generate` itself therefore cannot run and is not
    part of these vectors -- `IndexBasedLogitsProcessor` needs none of them.

Floats are stored as C99 hex strings (float.hex()) so that the comparison is exact.

  python tests/golden/make_reference_golden.py      # needs /root/reference; rewrites the json files next to it
"""
import json
import os
import random
import sys
import types

# The reference keeps the query n-grams of add_query_to_keys in a python set of STRINGS (retrieval.py:115-131): their order,
# and with it the order of the keys in ref_searcher.json, follows the interpreter's string hashing, which is randomised per
# process.  Pin it, so that re-running this script reproduces the committed files byte for byte.
if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("SEAL_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def _install_stand_ins():
    import torch  # noqa: F401
    import transformers

    mi = types.ModuleType("more_itertools")

    def chunked(iterable, n):
        buf = []
        for x in iterable:
            buf.append(x)
            if len(buf) == n:
                yield buf
                buf = []
        if buf:
            yield buf
    mi.chunked = chunked
    mi.ichunked = chunked
    sys.modules["more_itertools"] = mi

    # resolve the lazily imported names the reference needs FIRST: resolving them later swaps the
    # package's module object and with it anything set from outside
    from transformers import (AutoConfig, AutoModelForSeq2SeqLM, AutoTokenizer, BartForConditionalGeneration,  # noqa: F401
                              BartTokenizer, LogitsProcessor, LogitsProcessorList, StoppingCriteriaList)
    transformers = sys.modules["transformers"]          # importing the torch models REPLACES the package's module object
    for name in ("BeamScorer", "BeamSearchScorer", "HammingDiversityLogitsProcessor"):
        if name not in transformers.__dict__:
            setattr(transformers, name, type(name, (), {}))
    gu = types.ModuleType("transformers.generation_utils")
    for name in ("BeamSearchOutput", "validate_stopping_criteria", "BeamSearchEncoderDecoderOutput", "BeamSearchDecoderOnlyOutput"):
        setattr(gu, name, object)
    sys.modules["transformers.generation_utils"] = gu
    glp = types.ModuleType("transformers.generation_logits_process")
    glp.TopKLogitsWarper = object
    sys.modules["transformers.generation_logits_process"] = glp

    import numpy as np
    from oracle.seal_oracle import CppFMIndex

    class SwigLikeFMIndex(CppFMIndex):
        """the C++ initialize_from_file builds the index itself; it must not dispatch to the Python
        subclass's initialize() the way a plain Python call would"""
        def initialize_from_file(self, path, width):
            dt = {1: "<u1", 2: "<u2", 4: "<u4", 8: "<u8"}[width]
            CppFMIndex.initialize(self, np.fromfile(path, dtype=dt).astype(np.uint64))

    shim = types.ModuleType("seal.cpp_modules.fm_index")
    shim.FMIndex = SwigLikeFMIndex

    def load_FMIndex(path):
        raise NotImplementedError("the stand-in index is built in memory")
    shim.load_FMIndex = load_FMIndex
    sys.modules["seal.cpp_modules.fm_index"] = shim
    sys.path.insert(0, REFERENCE)


def fhex(x):
    return float(x).hex()


def make_docs(rng, n_docs, vocab, min_len=5, max_len=18, title_sep=7):
    """documents shaped like SEAL's: title tokens, the title delimiter, body, eos (2); a few repeated
    bodies so that keys occur in several documents and inside one document more than once"""
    docs = []
    for _ in range(n_docs):
        title = [rng.randrange(8, vocab) for _ in range(rng.randrange(1, 4))]
        body = [rng.randrange(8, vocab) for _ in range(rng.randrange(min_len, max_len))]
        if docs and rng.random() < 0.3:
            src = rng.choice(docs)
            a = rng.randrange(0, max(1, len(src) - 4))
            body[2:2] = src[a:a + rng.randrange(2, 5)]
        if rng.random() < 0.2:
            body += body[:3]
        docs.append(title + [title_sep] + body + [2])
    docs.append([9, 9, 9, 9, 9, 9, 9, 2])             # windows of one key overlapping each other
    return docs


def make_keys(rng, docs, vocab, n_keys):
    keys = []
    for _ in range(n_keys):
        d = rng.choice(docs)
        a = rng.randrange(0, len(d) - 1)
        ng = d[a:a + rng.randrange(1, 5)]
        r = rng.random()
        if r < 0.12:
            ng = ng + [rng.randrange(8, vocab)]       # mostly absent from the corpus
        elif r < 0.25:
            ng = [2] + d[:3]                          # title keys start with the previous document's eos
        keys.append((list(ng), -(rng.random() * 6 + 0.05)))
    keys += [([9, 9], -0.7), ([9, 9, 9], -1.1)]
    if rng.random() < 0.5:
        keys += [keys[0], (keys[1][0], keys[1][1] - 0.3)]
    return keys


def dump_results(results):
    out = []
    for doc, info in results.items():
        out.append({"doc": int(doc), "score": fhex(info[0]),
                    "keys": [[list(map(int, k)), fhex(s)] for k, s in info[1]],
                    "tokens": [int(t) for t in info[3]],
                    "best": [list(map(int, info[4][0])), fhex(info[4][1])]})
    return out


def aggregate_cases():
    from seal.index import FMIndex
    from seal.keys import aggregate_evidence
    option_space = dict(
        with_unigrams=[False, True], mode=["score", "length", "freq"], allow_overlaps=[False, True], single_key=[0.0, 0.3, 1.0],
        use_fm_index_frequency=[True, False], add_best_unigrams_to_ngrams=[False, True], single_key_add_unigrams=[False, True],
        unigrams_ignore_free_places=[False, True], max_occurrences_1=[3, 20, 1500], n_docs_complete_score=[3, 8, 500],
        use_top_k_unigrams=[0, 5, 1000], beta=[0.0, 0.8, 1.0], alpha=[1.0, 2.0], length_penalty=[0.0, 0.2], smoothing=[5.0, 0.5])
    pick = random.Random(20260925)
    cases = []
    for it in range(36):
        kw = {k: pick.choice(v) for k, v in option_space.items()}
        if it == 0:                                    # the searcher's own defaults (retrieval.py:401-446)
            kw.update(with_unigrams=True, mode="score", allow_overlaps=False, single_key=0.0, use_fm_index_frequency=True,
                      add_best_unigrams_to_ngrams=True, single_key_add_unigrams=False, unigrams_ignore_free_places=False,
                      max_occurrences_1=1500, n_docs_complete_score=1500, use_top_k_unigrams=5000, beta=0.8, alpha=2.0,
                      length_penalty=0.0, smoothing=5.0)
        mode = kw.pop("mode")
        kw["sort_by_length"], kw["sort_by_freq"] = mode == "length", mode == "freq"
        with_unigrams = kw.pop("with_unigrams")
        rng = random.Random(5000 + it)
        vocab, n_docs = rng.choice([14, 40, 70]), rng.choice([6, 25, 50])
        docs = make_docs(rng, n_docs, vocab)
        index = FMIndex()
        index.initialize(docs, in_memory=True)
        keys = make_keys(rng, docs, vocab, rng.randrange(5, 40))
        us = None
        if with_unigrams:
            us = [-(rng.random() * 9 + 0.05) for _ in range(vocab + 12)]
            if it % 3 == 0:
                us = [float(min(round(x), -1)) for x in us]
        results, all_ngrams = aggregate_evidence([(list(k), s) for k, s in keys], unigram_scores=None if us is None else list(us),
                                                 index=index, **kw)
        cases.append({"docs": docs, "keys": [[k, fhex(s)] for k, s in keys], "unigram_scores": None if us is None else [fhex(x) for x in us],
                      "kwargs": kw, "results": dump_results(results),
                      "all_ngrams": [[list(map(int, k)), fhex(s)] for k, s in all_ngrams.items()]})
    return cases


def _split_words(query):
    """stands in for spaCy's English tokenizer (absent offline): whitespace words as objects with .text"""
    from tests.helpers import split_words
    return split_words(query)


def helper_cases():
    from seal.keys import decompose_query_into_keys, deduplicate, strip
    rng = random.Random(77)
    out = {"strip": [], "deduplicate": [], "decompose": []}
    for q in ("who wrote the hobbit", " a ", "x", "", "New york city marathon 2019 winner", "a b"):
        for length in (1, 3):
            out["decompose"].append({"query": q, "length": length, "keys_sorted": sorted(decompose_query_into_keys(q, _split_words, length))})
    for _ in range(60):
        seq = [rng.randrange(0, 6) for _ in range(rng.randrange(0, 9))]
        st, en = sorted(rng.sample(range(6), 2)), sorted(rng.sample(range(6), 2))
        out["strip"].append({"seq": seq, "start": st, "end": en, "out": strip(list(seq), st, en)})
    for _ in range(30):
        items = []
        for _ in range(rng.randrange(1, 10)):
            k = [rng.randrange(0, 4) for _ in range(rng.randrange(1, 4))]
            items.append([-rng.random(), k] if rng.random() < 0.5 else k)
        typed = [(x[0], x[1]) if isinstance(x[0], float) else x for x in items]
        kept = deduplicate(typed)
        out["deduplicate"].append({"items": items, "kept_positions": [next(i for i, y in enumerate(typed) if y is x) for x in kept]})
    return out


def index_and_processor_cases():
    import torch
    from seal.beam_search import IndexBasedLogitsProcessor
    from seal.index import FMIndex
    cases = []
    for it in range(8):
        rng = random.Random(9000 + it)
        vocab = rng.choice([30, 60])
        docs = make_docs(rng, rng.choice([10, 40]), vocab - 8)
        index = FMIndex()
        index.initialize(docs, in_memory=it % 2 == 0)           # both branches of index.py:39-66
        n = index.size()
        case = {"docs": docs, "vocab": vocab, "size": n, "len": len(index), "n_docs": index.n_docs, "beginnings": list(index.beginnings),
                "occurring_sorted": sorted(index.occurring), "occurring_distinct": list(index.occurring_distinct),
                "occurring_counts": list(index.occurring_counts)}
        seqs = []
        for _ in range(40):
            d = rng.choice(docs)
            a = rng.randrange(len(d))
            s = d[a:a + rng.randrange(1, 6)]
            if rng.random() < 0.2:
                s = s + [rng.randrange(3, vocab)]
            seqs.append(s)
        seqs += [[], [vocab + 5], [2], [2, 2]]
        case["ranges"] = [[s, list(index.get_range(list(s))), index.get_count(list(s))] for s in seqs]
        case["continuations"] = [[s, sorted(index.get_continuations(list(s)))] for s in seqs[:15]]
        rows = [rng.randrange(n) for _ in range(30)]
        case["rows"] = [[r, index.get_token_index_from_row(r), index.get_doc_index_from_row(r)] for r in rows]
        case["docs_back"] = [index.get_doc(i) for i in range(index.n_docs)]
        lows = [rng.randrange(0, n) for _ in range(12)]
        spans = [(lo, min(n, lo + rng.randrange(0, 40))) for lo in lows] + [(0, n), (0, n - 1), (3, 3)]
        case["distinct_count"] = [[lo, hi, [list(x) for x in index.get_distinct_count(lo, hi)]] for lo, hi in spans]
        multi = index.get_distinct_count_multi([lo for lo, _ in spans], [hi for _, hi in spans])
        case["distinct_count_multi_equals_single"] = [[list(a), list(b)] for a, b in multi] == [c[2] for c in case["distinct_count"]]

        # IndexBasedLogitsProcessor.__call__ (beam_search.py:62-140): the additive mask, as the set of allowed tokens per row
        masks = []
        for kw in (dict(), dict(force_decoding_from=[2], eos_token_id=7), dict(stop_at_count=2), dict(always_allow_eos=True),
                   dict(forced_bos_token_id=5)):
            eos = kw.get("eos_token_id", 2)
            proc = IndexBasedLogitsProcessor(index, 4, pad_token_id=1, eos_token_id=eos,
                                             **{k: v for k, v in kw.items() if k != "eos_token_id"})
            for cur_len in (1, 2, 3, 5):
                rows_ids = []
                for i in range(8):
                    d = rng.choice(docs)
                    a = 0 if kw.get("force_decoding_from") else rng.randrange(len(d))
                    sent = ([2] + d[a:a + cur_len - 1] + [1] * cur_len)[:cur_len]
                    if i % 4 == 3 and cur_len > 2:
                        sent[-1] = rng.randrange(8, vocab - 8)            # usually leaves the corpus
                    if i % 8 == 6 and cur_len > 1:
                        sent[-1] = eos
                    rows_ids.append(sent)
                ids = torch.tensor(rows_ids)
                scores = torch.zeros(len(rows_ids), vocab)
                out = proc(ids, scores)
                allowed = [[t for t in range(vocab) if out[r, t].item() == 0.0] for r in range(len(rows_ids))]
                assert all(v == 0.0 or v == float("-inf") for v in out.flatten().tolist())
                masks.append({"kwargs": kw, "input_ids": rows_ids, "allowed": allowed})
        case["masks"] = masks
        cases.append(case)
    return cases


def model_cases():
    """compute_unigram_scores (keys.py:145-176) and rescore_keys (keys.py:64-141) with a seeded tiny BART on CPU.
    rescore_keys calls a private HF hook whose signature changed since; it is given back its old form
    (encode the inputs once) -- nothing else of the function is touched."""
    import torch
    from seal.keys import compute_unigram_scores, rescore_keys
    from tests.helpers import tiny_bart
    vocab = 120
    model = tiny_bart(vocab=vocab, seed=3)

    def old_prepare(input_ids, model_kwargs, *a, **k):
        enc = model.get_encoder()(input_ids=input_ids, attention_mask=model_kwargs["attention_mask"], return_dict=True)
        return {**model_kwargs, "encoder_outputs": enc}
    model._prepare_encoder_decoder_kwargs_for_generation = old_prepare
    rng = random.Random(4242)
    inputs = [[0] + [rng.randrange(4, vocab) for _ in range(rng.randrange(3, 9))] + [2] for _ in range(5)]
    out = {"vocab": vocab, "model_seed": 3, "inputs": inputs}
    with torch.no_grad():
        out["unigram_scores"] = [[fhex(x) for x in row] for row in compute_unigram_scores(model, inputs, None, tolist=True)]
        out["unigram_scores_prefix"] = [[fhex(x) for x in row] for row in compute_unigram_scores(model, inputs, None, tolist=True, prefix=[5, 9])]
        decoded = []
        for _ in inputs:
            keys = []
            for _ in range(rng.randrange(2, 7)):
                k = [rng.randrange(4, vocab) for _ in range(rng.randrange(1, 6))]
                if rng.random() < 0.4:
                    k = [2] + k
                if rng.random() < 0.4:
                    k = k + [rng.choice([2, 7])]
                keys.append((-rng.random(), k) if rng.random() < 0.5 else k)
            decoded.append(keys)
        out["decoded"] = [[[fhex(x[0]), x[1]] if isinstance(x, tuple) else x for x in kk] for kk in decoded]
        runs = []
        for kw in (dict(), dict(length_penalty=1.0), dict(strip_from_bos=[2, 0], strip_from_eos=[2, 7]), dict(prefix=[5])):
            res = rescore_keys(model, inputs, decoded, batch_size=4, **kw)
            runs.append({"kwargs": kw, "scores": [[[fhex(s), list(k)] for s, k in per_query] for per_query in res]})
        out["rescore"] = runs
    return out


class _ModelForTheReferenceLoop:
    """What seal/beam_search.py::fm_index_generate + constrained_beam_search need from a HF-4.1x model, for this
    configuration, given back by hand because transformers 5 no longer has these hooks: the logits processors
    HF would have built for these arguments (only InfNanRemove: min_length is disabled by eos_token_id=None,
    forced_bos is None), a max-length stopping criterion, encoder inputs expanded per beam, and a cache-free
    forward of the REAL seeded tiny BART.  The loop, the scorer with memory, the hypotheses and the
    constraint processor are the reference's own code."""

    def __init__(self, bart, enc_ids, enc_mask):
        self.bart, self.config = bart, bart.config
        self.enc_ids, self.enc_mask = enc_ids, enc_mask
        for name, default in (("output_scores", False), ("output_attentions", False), ("output_hidden_states", False),
                              ("return_dict_in_generate", False)):
            if getattr(self.config, name, None) is None:
                setattr(self.config, name, default)

    def _get_logits_processor(self, **kw):
        from transformers import InfNanRemoveLogitsProcessor, LogitsProcessorList
        assert kw["eos_token_id"] is None and kw["forced_bos_token_id"] is None and kw["remove_invalid_values"]
        return LogitsProcessorList([InfNanRemoveLogitsProcessor()])

    def _get_stopping_criteria(self, max_length, max_time):
        class MaxLength(list):
            def __init__(self, n):
                super().__init__([n])
                self.max_length = n

            def __call__(self, input_ids, scores):
                return input_ids.shape[-1] >= self.max_length
        return MaxLength(max_length)

    def _prepare_encoder_decoder_kwargs_for_generation(self, input_ids, kwargs):
        return dict(kwargs, encoder_outputs=None)

    def _prepare_decoder_input_ids_for_generation(self, batch_size, decoder_start_token_id, bos_token_id):
        import torch
        return torch.full((batch_size, 1), decoder_start_token_id, dtype=torch.long)

    def _expand_inputs_for_generation(self, decoder_input_ids, expand_size, is_encoder_decoder, **model_kwargs):
        self.rep_ids = self.enc_ids.repeat_interleave(expand_size, 0)
        self.rep_mask = self.enc_mask.repeat_interleave(expand_size, 0)
        model_kwargs.pop("attention_mask", None)
        return decoder_input_ids.repeat_interleave(expand_size, 0), model_kwargs

    def prepare_inputs_for_generation(self, input_ids, **kw):
        return {"decoder_input_ids": input_ids}

    def __call__(self, decoder_input_ids, return_dict=True, output_attentions=None, output_hidden_states=None):
        return self.bart(input_ids=self.rep_ids, attention_mask=self.rep_mask, decoder_input_ids=decoder_input_ids)

    def adjust_logits_during_generation(self, logits, cur_len):
        return logits

    def _update_model_kwargs_for_generation(self, outputs, model_kwargs, is_encoder_decoder):
        return dict(model_kwargs, past=None)


def beam_cases(model_kw=None, configurations=None):
    """seal/beam_search.py::fm_index_generate(keep_history=True) -- the reference's whole decode: constrained beam
    loop, BeamSearchScorerWithMemory, BeamHypothesesWithMemory, IndexBasedLogitsProcessor.
    ``model_kw``: another geometry of the seeded tiny BART (d_model=128, heads=2: head_dim 64, the one the fused
    sealnn_* decoder kernels run at -- the *_dh64.json fixtures of the -m gpu twins); ``configurations``: which of
    the seven option sets"""
    import torch
    from seal.beam_search import fm_index_generate
    from seal.index import FMIndex
    from tests.helpers import tiny_bart
    vocab = 120
    bart = tiny_bart(vocab, **(model_kw or {}))
    rng = random.Random(31)
    docs = []
    for _ in range(150):
        toks = [rng.randrange(4, vocab) for _ in range(rng.randrange(3, 15))]
        docs.append(toks[:2] + [7] + toks[2:] + [2])
    index = FMIndex()
    index.initialize(docs, in_memory=True)
    torch.manual_seed(2)
    enc_ids = torch.randint(4, vocab, (4, 8))
    enc_mask = torch.ones_like(enc_ids)
    cases = []
    option_sets = (dict(max_length=6, num_beams=3, length_penalty=0.0),
                   dict(max_length=5, num_beams=4, length_penalty=1.0),
                   dict(max_length=7, num_beams=3, length_penalty=0.0, force_decoding_from=[2], eos_token_id=7),
                   dict(max_length=5, num_beams=2, length_penalty=0.0, always_allow_eos=True),
                   dict(max_length=5, num_beams=3, length_penalty=0.0, stop_at_count=2),
                   dict(max_length=8, num_beams=5, length_penalty=0.5, stop_at_count=1, always_allow_eos=True),
                   dict(max_length=4, num_beams=3, length_penalty=0.0, disable_fm_index=True))
    for kw in (option_sets if configurations is None else [option_sets[i] for i in configurations]):
        model = _ModelForTheReferenceLoop(bart, enc_ids, enc_mask)
        with torch.no_grad():
            out = fm_index_generate(model, index, enc_ids, enc_mask, min_length=1, keep_history=True, **kw)
        cases.append({"kwargs": kw, "hypotheses": [[[fhex(s), [int(t) for t in toks]] for s, toks in per_query] for per_query in out]})
    return {"vocab": vocab, **({"model_kw": model_kw} if model_kw else {}), "docs": docs, "enc_ids": enc_ids.tolist(), "cases": cases}


class _ModelForTheReferenceSearcher(_ModelForTheReferenceLoop):
    """the same hand-made HF-4.1x hooks, for a model that is handed encoder inputs per call (process_batch,
    rescore_keys and compute_unigram_scores all use it)"""

    def __init__(self, bart):
        import torch
        super().__init__(bart, torch.zeros(1, 1, dtype=torch.long), torch.ones(1, 1, dtype=torch.long))

    def parameters(self):
        return self.bart.parameters()

    def _prepare_encoder_decoder_kwargs_for_generation(self, input_ids, kwargs):
        self.enc_ids, self.enc_mask = input_ids, kwargs["attention_mask"]
        enc = self.bart.get_encoder()(input_ids=input_ids, attention_mask=kwargs["attention_mask"], return_dict=True)
        return dict(kwargs, encoder_outputs=enc)

    def _expand_inputs_for_generation(self, decoder_input_ids, expand_size, is_encoder_decoder, **model_kwargs):
        model_kwargs.pop("encoder_outputs", None)
        return super()._expand_inputs_for_generation(decoder_input_ids, expand_size, is_encoder_decoder, **model_kwargs)

    def __call__(self, decoder_input_ids=None, return_dict=True, output_attentions=None, output_hidden_states=None, **kw):
        if "input_ids" in kw:                      # rescore_keys / compute_unigram_scores: a plain forward
            return self.bart(decoder_input_ids=decoder_input_ids, **kw)
        return super().__call__(decoder_input_ids)


def searcher_cases(model_kw=None, runs=None):
    """(``model_kw`` / ``runs``: as in ``beam_cases`` -- another model geometry, a subset of the four runs.)
    seal/retrieval.py end to end: SEALSearcher.batch_search = batch_generate_keys.process_batch (body decode,
    post-filters, rescoring, title decode, title filters, rescoring, dedup, unigram scores) + retrieve_from_keys
    (aggregate_evidence with the searcher's parameters) + SEALDocument.  Two runs without the query n-gram keys and one
    with them (add_query_to_keys=True, the reference's default: retrieval.py:115-149 with a whitespace word tokenizer in
    spaCy's place and tests.helpers.ToyTokenizer as the BART tokenizer); the 'bart' backbone's hard-wired title / code
    delimiter ids are re-pointed at this toy vocabulary."""
    import numpy as np
    import torch
    import seal.retrieval as ref_retrieval
    from seal.index import FMIndex
    from seal.retrieval import SEALSearcher
    from tests.helpers import ToyTokenizer, make_docs as helper_docs, tiny_bart
    vocab, K, length, title_eos = 120, 4, 6, 7
    docs = helper_docs(5, 200, vocab - 8, min_len=6, max_len=18, title_sep=title_eos)
    index = FMIndex()
    index.initialize(docs, in_memory=True)
    index.labels = [f"d{i}" for i in range(len(docs))]
    rng = np.random.default_rng(0)
    queries_ids = [[0] + rng.integers(4, vocab - 8, size=int(rng.integers(4, 9))).tolist() + [2] for _ in range(3)]
    queries = [" ".join(f"w{t}" for t in q[1:-1]) for q in queries_ids]
    real_generate = ref_retrieval.fm_index_generate
    out = {"vocab": vocab, **({"model_kw": model_kw} if model_kw else {}), "beam": K, "length": length, "title_eos": title_eos, "docs": docs,
           "queries": queries_ids, "runs": []}
    ref_retrieval.word_tokenizer = _split_words
    # 15 is the reference's constant; 8 is what tests/test_gpu_search.py runs both sides at
    # the fourth run switches the code decode on (retrieval.py:212-264; partial_code: this corpus has no code sections, so
    # complete code keys do not exist)
    all_runs = ((8, False, False), (15, False, False), (8, True, False), (8, False, True))
    for title_length, query_keys, code in (all_runs if runs is None else [all_runs[i] for i in runs]):
        def generate(*a, **kw):
            if kw.get("force_decoding_from"):
                kw = {**kw, "max_length": title_length}
            return real_generate(*a, **kw)
        ref_retrieval.fm_index_generate = generate
        try:
            s = SEALSearcher(index, ToyTokenizer(vocab), _ModelForTheReferenceSearcher(tiny_bart(vocab, **(model_kw or {}))), backbone="bart-tiny",
                             length=length, beam=K, batch_size=2, add_query_to_keys=query_keys, detokenize=True,
                             **(dict(decode_code=True, partial_code=True) if code else {}))
            # (include_keys=True cannot be used with more than one query: batch_search's `for k, _ in kk` rebinds its own
            #  parameter k, the islice stop; the per-document keys are read from retrieve_from_keys below instead)
            s.title_eos_token_id, s.code_bos_token_id, s.code_eos_token_id = title_eos, title_eos, vocab - 6
            with torch.no_grad():
                keys = list(s.batch_generate_keys(queries))
                evidence = [s.retrieve_from_keys(kk) for kk in keys]
                retrieved = s.batch_search(queries, k=10)
        finally:
            ref_retrieval.fm_index_generate = real_generate
        run = {"title_length": title_length, "add_query_to_keys": query_keys, "queries": []}
        if code:
            run["decode_code"] = True
        for (kk, us), (res, _), docs_q in zip(keys, evidence, retrieved):
            assert [d.idx for d in docs_q] == list(res)[:10] and [d.score for d in docs_q] == [res[d.idx][0] for d in docs_q]
            run["queries"].append({
                "keys": [[list(map(int, n)), fhex(sc)] for n, sc in kk],
                "unigram_scores": [fhex(x) for x in us],
                "evidence": dump_results(dict(list(res.items())[:10])),
                "ranked": [{"doc": int(d.idx), "docid": d.docid, "score": fhex(d.score), "title": d.text()[0], "body": d.text()[1],
                            "raw_tokens": [int(t) for t in d._raw_tokens]} for d in docs_q]})
        out["runs"].append(run)
    return out


def checkpoint_cases():
    """seal/utils.py::load_state_dict_from_fairseq_checkpoint on a synthetic checkpoint (tests.helpers): checksums of
    every tensor of the loaded model and its logits on a fixed input"""
    import tempfile
    import torch
    from seal.utils import load_state_dict_from_fairseq_checkpoint
    from tests.helpers import synthetic_fairseq_checkpoint, tiny_bart
    vocab = 120
    model = tiny_bart(vocab, seed=99)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ckpt.pt")
        synthetic_fairseq_checkpoint(path, vocab=vocab, seed=11)
        load_state_dict_from_fairseq_checkpoint(model, path)
    model.eval()
    state = {k: [list(v.shape), fhex(v.double().sum().item()), fhex(v.double().abs().sum().item())] for k, v in model.state_dict().items()}
    enc = torch.tensor([[0, 5, 17, 33, 2], [0, 9, 9, 2, 1]])
    with torch.no_grad():
        logits = model(input_ids=enc, attention_mask=(enc != 1).long(), decoder_input_ids=torch.tensor([[2, 5], [2, 9]])).logits
    return {"vocab": vocab, "target_seed": 99, "checkpoint_seed": 11, "state": state,
            "logits_sample": [fhex(x) for x in logits[:, -1, :16].flatten().tolist()],
            "lm_head_is_embedding": bool(torch.equal(model.lm_head.weight, model.model.shared.weight))}


def main():
    _install_stand_ins()
    with open(os.path.join(HERE, "ref_checkpoint.json"), "w") as f:
        json.dump({"source": "seal/utils.py::load_state_dict_from_fairseq_checkpoint", **checkpoint_cases()}, f)
    with open(os.path.join(HERE, "ref_searcher.json"), "w") as f:
        json.dump({"source": "seal/retrieval.py::SEALSearcher.batch_search on tests.helpers.tiny_bart(120), CPU fp32", **searcher_cases()}, f)
    with open(os.path.join(HERE, "ref_beam_search.json"), "w") as f:
        json.dump({"source": "seal/beam_search.py::fm_index_generate on tests.helpers.tiny_bart(120), CPU fp32, HF-4.1x hooks given back by hand",
                   **beam_cases()}, f)
    # the same two, at the head_dim-64 geometry the fused decoder kernels run at (tests/test_gpu_reference_golden.py holds the HIP index +
    # fused step decoder to them directly)
    dh64 = dict(d_model=128, heads=2)
    with open(os.path.join(HERE, "ref_searcher_dh64.json"), "w") as f:
        json.dump({"source": "seal/retrieval.py::SEALSearcher.batch_search on tests.helpers.tiny_bart(120, d_model=128, heads=2), CPU fp32",
                   **searcher_cases(model_kw=dh64, runs=(0, 2))}, f)
    with open(os.path.join(HERE, "ref_beam_search_dh64.json"), "w") as f:
        json.dump({"source": "seal/beam_search.py::fm_index_generate on tests.helpers.tiny_bart(120, d_model=128, heads=2), CPU fp32, HF-4.1x hooks "
                             "given back by hand", **beam_cases(model_kw=dh64, configurations=(0, 1, 2, 4, 5))}, f)
    with open(os.path.join(HERE, "ref_model_side.json"), "w") as f:
        json.dump({"source": "seal/keys.py::compute_unigram_scores, rescore_keys on tests.helpers.tiny_bart(vocab=120, seed=3), CPU fp32",
                   **model_cases()}, f)
    with open(os.path.join(HERE, "ref_aggregate_evidence.json"), "w") as f:
        json.dump({"source": "seal/keys.py::aggregate_evidence run by tests/golden/make_reference_golden.py", "cases": aggregate_cases()}, f)
    with open(os.path.join(HERE, "ref_helpers.json"), "w") as f:
        json.dump({"source": "seal/keys.py::strip, deduplicate", **helper_cases()}, f)
    with open(os.path.join(HERE, "ref_index_and_mask.json"), "w") as f:
        json.dump({"source": "seal/index.py::FMIndex and seal/beam_search.py::IndexBasedLogitsProcessor.__call__",
                   "cases": index_and_processor_cases()}, f)
    for name in ("ref_aggregate_evidence.json", "ref_helpers.json", "ref_index_and_mask.json", "ref_model_side.json", "ref_beam_search.json", "ref_searcher.json", "ref_checkpoint.json",
                 "ref_beam_search_dh64.json", "ref_searcher_dh64.json"):
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__":
    main()
