"""Shared test fixtures: tiny seeded BART + corpora + oracle-backed processor."""
import numpy as np
import torch


def tiny_bart(vocab=120, seed=0, d_model=32, layers=2, heads=4, max_positions=64):
    """seeded random-init BART.  ``d_model=128, heads=2`` gives head_dim 64 = BART-large's, the geometry
    the fused ``sealnn_*`` decoder kernels are built for (seal_amd/bart_decoder.py)."""
    from transformers import BartConfig, BartForConditionalGeneration
    cfg = BartConfig(vocab_size=vocab, d_model=d_model, encoder_layers=layers, decoder_layers=layers,
                     encoder_attention_heads=heads, decoder_attention_heads=heads, encoder_ffn_dim=2 * d_model,
                     decoder_ffn_dim=2 * d_model, max_position_embeddings=max_positions)
    cfg.forced_bos_token_id = None       # as SEALSearcher.load_bart does (reference retrieval.py:566,580)
    torch.manual_seed(seed)
    m = BartForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        m.final_logits_bias[0, cfg.pad_token_id] = float("-inf")   # reference retrieval.py:584-588
        m.final_logits_bias[0, cfg.bos_token_id] = float("-inf")
    return m


def make_docs(seed, n_docs, vocab, min_len=3, max_len=14, title_sep=None):
    rng = np.random.default_rng(seed)
    docs = []
    for _ in range(n_docs):
        m = int(rng.integers(min_len, max_len + 1))
        toks = rng.integers(4, vocab, size=m).tolist()
        if title_sep is not None:
            toks = toks[:2] + [title_sep] + toks[2:]
        docs.append(toks + [2])
    return docs


# head_dim 8 (torch fallback of the step decoder) and head_dim 64 (fused sealnn_* kernels, as BART-large)
MODEL_GEOMETRIES = [dict(d_model=32, heads=4), dict(d_model=128, heads=2)]
MODEL_IDS = ["dh8-fallback", "dh64-fused"]


class OracleLogitsProcessor:
    """tests-only: IndexBasedLogitsProcessor protocol backed by the CPU oracle, so the
    tensorised beam loop (host logic) can be exercised without a GPU."""

    def __init__(self, index, num_beams, vocab, **kw):
        self.index, self.num_beams, self.vocab, self.kw = index, num_beams, vocab, kw

    def __call__(self, input_ids, scores):
        from oracle.beam_oracle import oracle_logits_mask
        allowed = oracle_logits_mask(self.index, input_ids.tolist(), self.vocab, self.num_beams, **self.kw)
        mask = torch.full_like(scores, float("-inf"))
        mask[torch.from_numpy(allowed).to(scores.device)] = 0.0
        return scores + mask


def hf_logits_fn(model, enc_ids, enc_mask, num_beams):
    """next-token logits through HF's own full (cache-free) forward."""
    ids = enc_ids.repeat_interleave(num_beams, 0)
    am = enc_mask.repeat_interleave(num_beams, 0)

    def fn(decoder_input_ids):
        with torch.no_grad():
            return model(input_ids=ids, attention_mask=am, decoder_input_ids=decoder_input_ids.to(ids.device)).logits[:, -1, :]
    return fn


def valid_set(hyps, index, title_eos=None):
    """What the searcher keeps of a hypothesis list: strip, ``count > 0`` filter
    (reference retrieval.py:85-91), then first-occurrence dedup (retrieval.py:281).
    This is the level at which the reference itself is deterministic: when a query
    has fewer than 2K finite constrained candidates, ``torch.topk`` picks ``-inf``
    entries in an unspecified order; such picks either fail the count
    filter or (a stray eos/pad after a live prefix) strip back to a key that the
    live prefix already contributed at the previous step."""
    out = {}
    if title_eos is not None:
        # title decode: the searcher's title filter instead (retrieval.py:178-191): must end
        # with the title eos, keeps the leading </s>, count > 0 on the whole thing
        from oracle.keys_oracle import oracle_title_postfilter
        for score, k in oracle_title_postfilter(hyps, index, title_bos=2, title_eos=title_eos):
            out.setdefault(tuple(k), []).append(score)
        return {k: v[:1] for k, v in out.items()}
    for score, toks in hyps:
        k = list(toks)
        for _ in range(2):
            if k and k[0] in (0, 2):
                k = k[1:]
        if k and k[-1] in (0, 2):
            k = k[:-1]
        if k and index.get_count(k) > 0:
            out.setdefault(tuple(k), []).append(score)
    return {k: v[:1] for k, v in out.items()}


class OracleBatchIndex:
    """tests-only adapter: the batched index methods the host logic of
    seal_amd.keys consumes, answered by scalar calls into the CPU oracle."""

    def __init__(self, orc):
        self.orc = orc
        self.beginnings = orc.beginnings
        self.occurring_distinct = orc.occurring_distinct

    def __len__(self):
        return len(self.orc)

    def get_range_batch(self, seqs):
        r = [self.orc.get_range(list(s)) for s in seqs]
        return np.asarray([a for a, _ in r], dtype=np.uint64), np.asarray([b for _, b in r], dtype=np.uint64)

    def get_count_batch(self, seqs):
        lo, hi = self.get_range_batch(seqs)
        return (hi - lo).astype(np.int64)

    def get_count(self, seq):
        return self.orc.get_count(list(seq))

    def locate_ranges(self, lows, highs, max_per_range):
        pos, doc, offs = [], [], [0]
        for a, b in zip(lows, highs):
            rows = list(range(int(a), int(b)))[:max_per_range]
            for r in rows:
                p = self.orc.locate(r)
                pos.append(p)
                doc.append(self.orc.get_doc_index(p))
            offs.append(len(pos))
        return np.asarray(pos, dtype=np.int64), np.asarray(doc, dtype=np.int64), np.asarray(offs, dtype=np.int64)

    def get_docs_batch(self, docs):
        return [self.orc.get_doc(int(d)) for d in docs]


def synthetic_keys(rng, docs, vocab, n_keys=40, with_titles=False):
    """(ngram, lm_logprob) pairs shaped like the searcher's output: corpus n-grams
    (some repeated inside documents), a few absent ones, a few unigrams."""
    keys = []
    for _ in range(n_keys):
        d = docs[int(rng.integers(len(docs)))]
        a = int(rng.integers(0, len(d) - 1))
        ng = d[a:a + int(rng.integers(1, 5))]
        if rng.random() < 0.15:
            ng = ng + [int(rng.integers(4, vocab))]
        if with_titles and rng.random() < 0.2:
            ng = [2] + d[:3]
        keys.append((list(ng), -float(rng.random() * 6 + 0.05)))
    return keys


def synthetic_fairseq_checkpoint(path, vocab=120, seed=11):
    """a fairseq-style BART checkpoint of tiny_bart geometry: BartModel key names without the "model." prefix,
    the shared embedding stored three times with ONE ROW FEWER than the HF model has (the reference loader
    appends a zero row, seal/utils.py:44-46), plus the bookkeeping entries it drops"""
    src = tiny_bart(vocab, seed=seed)
    sd = {k: v.clone() for k, v in src.model.state_dict().items()}
    emb = sd["shared.weight"][:-1].clone()
    for k in ("shared.weight", "encoder.embed_tokens.weight", "decoder.embed_tokens.weight"):
        sd[k] = emb.clone()
    del sd["shared.weight"]                      # fairseq has no such key; the loader creates it
    sd["encoder.version"] = torch.tensor([3.0])
    sd["decoder.version"] = torch.tensor([3.0])
    sd["decoder.output_projection.weight"] = emb.clone()
    torch.save({"model": sd}, path)


class ToyTokenizer:
    """"w17 w3 || body" <-> [0, 17, 3, V-2, V-3, 2]: enough of a tokenizer for retrieval.py's process_batch (pre-tokenised
    corpora have no text); markers get the ids the product's marker_token_ids are given in tests.  A capitalised word
    ("W17", reference keys.py:46-47) is the same token.  Shared by tests/golden/make_reference_golden.py (the
    reference's searcher) and the product's tests."""

    def __init__(self, vocab):
        self.special = {"||": vocab - 2, "body": vocab - 3, "title": vocab - 4, "+": vocab - 5, "code": vocab - 7}
        self.back = {v: k for k, v in self.special.items()}

    def _ids(self, text, add_special_tokens=True):
        ids = [self.special[t] if t in self.special else int(t[1:]) for t in text.split()]
        return [0] + ids + [2] if add_special_tokens else ids

    def __call__(self, texts, return_tensors=None, padding=False, truncation=False, add_special_tokens=True):
        rows = [self._ids(t, add_special_tokens) for t in texts]
        if return_tensors != "pt":
            return {"input_ids": rows}
        width = max(len(r) for r in rows)
        ids = torch.tensor([r + [1] * (width - len(r)) for r in rows])
        return {"input_ids": ids, "attention_mask": (ids != 1).long()}

    def decode(self, ids, skip_special_tokens=False, clean_up_tokenization_spaces=False):
        out = []
        for t in ids:
            t = int(t)
            if skip_special_tokens and t in (0, 1, 2):
                continue
            out.append(self.back.get(t, f"w{t}"))
        return " ".join(out)

    def batch_decode(self, seqs, **kw):
        return [self.decode(s, **kw) for s in seqs]

    def as_target_tokenizer(self):
        import contextlib
        return contextlib.nullcontext()


class Word:
    def __init__(self, text):
        self.text = text


def split_words(query):
    """stands in for spaCy's English tokenizer (absent offline): whitespace words as objects with .text"""
    return [Word(w) for w in query.split()]


import contextlib


@contextlib.contextmanager
def kernel_options(index, **opts):
    """forces launch-shape switches of the constraint / top-2K calls on ``index``'s handle (include/sealfm.h
    ``fmi_dev_set_option``: the environment is only read when a handle is created) and restores the built-in choices"""
    from seal_amd._lib import check, lib
    handle = getattr(index, "handle", index)
    for k, v in opts.items():
        check(lib().fmi_dev_set_option(handle, k.encode(), int(v)))
    try:
        yield
    finally:
        for k in opts:
            check(lib().fmi_dev_set_option(handle, k.encode(), -1))
