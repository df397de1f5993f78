"""``SEALSearcher`` / ``SEALDocument`` on the MI355X engine (reference seal/retrieval.py).

The searcher keeps the reference's surface -- ``DEFAULTS`` as the single source
of parameters (retrieval.py:401-446), ``set_params`` / ``add_args`` /
``from_args`` / ``load`` / ``search`` / ``batch_search`` / ``generate_keys`` /
``retrieve_from_keys`` / ``doc`` -- and the same key-generation recipe
(``process_batch``, retrieval.py:54-305).  Differences are confined to where
work runs:

* decoding goes through ``seal_amd.fm_index_generate`` (GPU constraint kernels);
* every ``fm_index.get_count(k) > 0`` post-filter of a batch (retrieval.py:91,
  130,191,247) is one batched backward-search launch;
* ``jobs`` (a ``multiprocessing.Pool`` over queries in the reference,
  retrieval.py:762-775): the per-query index work is batched on the GPU and the
  evidence aggregation runs there too (``seal_amd/csrc/fmi_aggregate.hip``);
  ``jobs >= 2`` only moves the bit-exact host checker routines
  (``fmi_first_stage`` / ``fmi_full_score``) into worker processes when the
  GPU aggregation is switched off (``gpu_aggregate=False``).

Queries may be strings (a HF tokenizer is then required, as in the reference)
or pre-tokenised id lists ``[<s>, ..., </s>]`` (no tokenizer needed; the marker
suffixes " || body", " || title", " || +" are then taken from
``marker_token_ids``).
"""
import logging
from itertools import islice
from typing import Dict, List, Optional, Sequence, Union

import torch

from . import keys as rk
from .beam_search import fm_index_generate, fm_index_generate_joint
from .index import FMIndex

TITLE_MAX_LENGTH = 15     # the reference hard-wires the title / code decode length (retrieval.py:165,215); tests shorten it

logger = logging.getLogger(__name__)

DEBUG = False


class SEALDocument:
    """A retrieved passage (reference retrieval.py:315-397)."""

    def __init__(self, idx: int, score: float, fm_index: FMIndex, bart_tokenizer, delim1: int = 49314,
                 delim2: int = None, keys=None, query=None):
        self.idx = idx
        self.score = score
        self.fm_index = fm_index
        self.bart_tokenizer = bart_tokenizer
        self.delim1 = delim1
        self.delim2 = delim2
        self.keys = keys
        self.query = query
        self._raw_tokens = None
        self._body = None
        self._title = None

    @property
    def docid(self):
        return self.fm_index.labels[self.idx]

    def id(self):
        return self.idx

    def raw_tokens(self):
        if self._raw_tokens is None:
            self._raw_tokens = self.fm_index.get_doc(self.idx)
        elif not isinstance(self._raw_tokens, list):
            self._raw_tokens = list(self._raw_tokens)        # the searcher hands the fetched token array over as it is
        return self._raw_tokens

    def raw_text(self):
        return self.bart_tokenizer.decode(self.raw_tokens(), clean_up_tokenization_spaces=False)

    def text(self):
        if self._body is None or self._title is None:
            title_tokens, body_tokens = self.split_tokens(self.raw_tokens())
            dec = lambda t: self.bart_tokenizer.decode(t, skip_special_tokens=True, clean_up_tokenization_spaces=False)
            self._title = dec(title_tokens) if title_tokens else ""
            self._body = dec(body_tokens)
        return self._title, self._body

    def split_tokens(self, tokens):
        """title = tokens before ``delim1`` ('@@'), body = after; ``delim2`` ('||')
        cuts a leading code section off the body (retrieval.py:368-394)."""
        if self.delim1 is None:
            title_tokens, body_tokens = [], []
        elif self.delim1 in tokens:
            i = tokens.index(self.delim1)
            title_tokens, body_tokens = tokens[:i], tokens[i + 1:]
        else:
            title_tokens, body_tokens = [], tokens
        i = 0
        if self.delim2 is not None and self.delim2 in body_tokens:
            i = body_tokens.index(self.delim2) + 1
        return title_tokens, body_tokens[i:]

    def __repr__(self):
        return f'<GRDocument: {self.idx}, "{self.raw_text()[:30]}[...]">'


def _chunks(it, size):
    it = iter(it)
    while True:
        block = list(islice(it, size))
        if not block:
            return
        yield block


def batch_generate_keys(searcher, queries, constrained_generation=True):
    """Generator of per-query keys (reference retrieval.py:49-312)."""
    offset = 0
    for batch in _chunks(queries, searcher.batch_size):
        for instance in _process_batch(searcher, batch, constrained_generation, offset):
            yield instance
        offset += len(batch)


# a caller that swaps this module's fm_index_generate for its own gets its own, i.e. no joint loop -- unless its stand-in
# says ``_joint_ok`` (bench.py's phase timer wraps both entry points)
_FM_INDEX_GENERATE = fm_index_generate


def _count_filter(index: FMIndex, per_query: List[List]) -> List[List]:
    """``[(s, k) for s, k in fk if k and index.get_count(k) > 0]`` for every query of
    the batch with one launch."""
    flat = [k for fk in per_query for _, k in fk if k]
    counts = iter(index.get_count_batch(flat).tolist() if flat else [])
    out = []
    for fk in per_query:
        kept = []
        for s, k in fk:
            if k and next(counts) > 0:
                kept.append((s, k))
        out.append(kept)
    return out


class _HypArrays:
    """The recorded hypotheses of one decode as arrays (``PendingGenerate.arrays``) with a window [start, end) per hypothesis:
    the reference's list comprehensions over (score, token list) pairs (retrieval.py:85-91, 180-191) become a few numpy
    operations over ~10^4 rows, and python lists are built for the keys that survive the filters only."""

    def __init__(self, arrays):
        import numpy as np
        tok, length, score, valid = arrays
        self.np = np
        self.nq, self.nh, self.L = tok.shape
        self.tok = tok.reshape(self.nq * self.nh, self.L)
        self.score = score.reshape(-1)
        self.keep = valid.reshape(-1).copy()
        self.start = np.zeros(self.nq * self.nh, dtype=np.int64)
        self.end = np.tile(length, self.nq)
        self.rows = np.arange(self.nq * self.nh)

    def _at(self, pos):
        return self.tok[self.rows, np_clip(pos, 0, self.L - 1)]

    def drop_empty(self):
        self.keep &= self.end > self.start

    def strip_front(self, ids):
        hit = self.keep & (self.end > self.start) & self.np.isin(self._at(self.start), ids)
        self.start += hit

    def strip_back(self, ids):
        hit = self.keep & (self.end > self.start) & self.np.isin(self._at(self.end - 1), ids)
        self.end -= hit

    def require_length(self, n):
        self.keep &= (self.end - self.start) == n

    def require_last(self, token):
        self.keep &= (self.end > self.start) & (self._at(self.end - 1) == token)

    def csr(self):
        """(indices of the kept, non-empty hypotheses in list order, offsets, flat tokens) for the count filter"""
        np = self.np
        idx = np.nonzero(self.keep & (self.end > self.start))[0]
        lens = (self.end - self.start)[idx]
        pos = np.arange(self.L)[None, :]
        m = (pos >= self.start[idx, None]) & (pos < self.end[idx, None])
        return idx, lens, self.tok[idx][m]

    def lists(self, idx, prepend=None):
        """per query ``[(score, token list)]`` of the hypotheses ``idx`` (ascending = the reference's list order); ``prepend``:
        a token put in front of keys that do not start with it (retrieval.py:190)"""
        out = [[] for _ in range(self.nq)]
        sc = self.score[idx].tolist()
        st, en = self.start[idx].tolist(), self.end[idx].tolist()
        rows = self.tok[idx].tolist()
        for i, s_, a, b, row in zip((idx // self.nh).tolist(), sc, st, en, rows):
            k = row[a:b]
            if prepend is not None and k[0] != prepend:
                k = [prepend] + k
            out[i].append((s_, k))
        return out


def np_clip(a, lo, hi):
    import numpy as np
    return np.clip(a, lo, hi)


def _process_batch(searcher, inputs, constrained_generation, offset=0):
    """keys of one batch of queries (reference retrieval.py:54-305)"""
    steps = _batch_steps(searcher, inputs, constrained_generation, offset)
    try:
        while True:
            next(steps)
    except StopIteration as done:
        return done.value


def _batch_steps(searcher, inputs, constrained_generation, offset=0):
    """``process_batch`` of the reference (retrieval.py:54-305) as a generator in segments, so that a scheduler can
    put other work between them (``SEALSearcher._overlapped_results``):

    1. the body decode is ENQUEUED -- nothing waits for the GPU -- ``yield "body"``; the title decode likewise, ``yield
       "decoding"``;
    2. the hypotheses come to the host (the one wait for the decodes), then ``yield "decoded"``;
    3. post-filters (one count launch and its read-back, no GEMM), ``yield "filtered"``; the rescorings enqueued, ``yield
       "rescoring"``; scores read back, query n-grams, unigram scores -> returns the keys (``StopIteration.value``).

    The reference runs body decode -> body filters/rescoring -> query keys -> title decode -> title filters/rescoring;
    the two decodes do not depend on anything in between, so issuing them back to back changes no result."""
    s = searcher
    fm_index = s.fm_index
    dec = lambda model: {}
    bias = s.logit_bias
    if bias is not None and bias.shape[0] != len(inputs):
        bias = bias[offset:offset + len(inputs)]      # one row per query of the whole call
    tokenised = not isinstance(inputs[0], str)
    if tokenised:
        base_tokens = [list(q) for q in inputs]
    else:
        inputs = [(" " + q.strip()) if s.prepend_space else q.strip() for q in inputs]
        base_tokens = s.bart_tokenizer(inputs, padding=False)["input_ids"]

    def marked(kind: str):
        """encoder inputs with ' || <kind>' and ' || +' appended (retrieval.py:60-65)"""
        if tokenised:
            extra = (s.marker_token_ids[kind] if s.use_markers else []) + (s.marker_token_ids["+"] if s.value_conditioning else [])
            return None, [t[:-1] + extra + t[-1:] for t in base_tokens]
        strs = inputs
        if s.use_markers:
            strs = [i + f" || {kind}" for i in strs]
        if s.value_conditioning:
            strs = [i + " || +" for i in strs]
        return strs, s.bart_tokenizer(strs, padding=False)["input_ids"]

    def encoder_batch(strs, toks):
        if strs is not None:
            b = s.bart_tokenizer(strs, return_tensors="pt", padding=True, truncation=True)
            return {k: v.to(s.device) for k, v in b.items()}
        ids = rk._pad_batch(toks, s.bart_model.config.pad_token_id, s.device)
        return dict(input_ids=ids, attention_mask=(ids != s.bart_model.config.pad_token_id).long())

    # `input_tokens` of the reference as its final rescoring sees it (retrieval.py:269-279 uses whatever was
    # bound last: the bare query, retrieval.py:57, unless the query-n-gram branch re-bound it, :139)
    last_input_tokens = base_tokens
    strip_ids = s.strip_token_ids
    bos_strip = [s.title_bos_token_id, s.code_bos_token_id, s.bart_model.config.decoder_start_token_id]

    # ---- segment 1: enqueue the decodes ----
    body = titles = None
    codes = code_toks = title_toks = None
    # One loop for the decodes of this batch that share a model (the default: ``bart_title_model is bart_model``): their rows
    # are stacked -- 2 x batch x beams per model step --, see ``fm_index_generate_joint``.  Needs the GPU index (one fused
    # constraint call per step serves every decode's rows) and the reference's settings for those decodes.
    joint_kinds = []
    if (getattr(s, "joint_decode", True) and s.device.type == "cuda" and hasattr(fm_index, "handle") and s.diverse_bs_groups == 1
            and not s.topk and getattr(fm_index_generate, "_joint_ok", fm_index_generate is _FM_INDEX_GENERATE)):
        joint_kinds = [k for k, on, m in (("body", s.decode_body, s.bart_model), ("title", s.decode_titles, s.bart_title_model),
                                          ("code", s.decode_code, s.bart_code_model)) if on and m is s.bart_model]
        if len(joint_kinds) < 2:
            joint_kinds = []
    if joint_kinds:
        # stop_at_count is the body decode's alone (reference retrieval.py:70-83 passes it; the title / code decodes, :162-176 and
        # :212-236, run with the default 0)
        job_of = {"body": dict(max_length=s.length, eos_token_id=None, force_decoding_from=None, stop_at_count=s.stop_at_count),
                  "title": dict(max_length=TITLE_MAX_LENGTH, eos_token_id=s.title_eos_token_id, force_decoding_from=[s.title_bos_token_id],
                                stop_at_count=0),
                  "code": dict(max_length=TITLE_MAX_LENGTH, eos_token_id=s.code_eos_token_id, force_decoding_from=[s.code_bos_token_id],
                               stop_at_count=0)}
        marked_in = {k: marked(k) for k in joint_kinds}
        # the bare queries -- what the rescoring of the body keys encodes (retrieval.py:93-100) -- ride along in the same encoder pass
        ride = s.decode_body and s.rescore and s.use_markers
        if tokenised:
            enc_in = encoder_batch(None, [t for k in joint_kinds for t in marked_in[k][1]] + (base_tokens if ride else []))
        else:
            enc_in = encoder_batch([x for k in joint_kinds for x in marked_in[k][0]] + (list(inputs) if ride else []), None)
        n_own = len(inputs) * len(joint_kinds)
        extra = (enc_in["input_ids"][n_own:], enc_in["attention_mask"][n_own:]) if ride else None
        enc_in = {k: v[:n_own] for k, v in enc_in.items()}
        pend = fm_index_generate_joint(
            s.bart_model, fm_index, enc_in["input_ids"], enc_in["attention_mask"],
            [dict(batch=len(inputs), **job_of[k]) for k in joint_kinds], extra_inputs=extra, num_beams=s.beam, length_penalty=s.length_penalty,
            stop_at_count=s.stop_at_count, disable_fm_index=not constrained_generation,
            logit_bias=torch.cat([bias] * len(joint_kinds)) if bias is not None else None, **dec(s.bart_model))
        got = dict(zip(joint_kinds, pend))
        body, titles, codes = got.get("body"), got.get("title"), got.get("code")
        title_toks = marked_in["title"][1] if "title" in got else None
        code_toks = marked_in["code"][1] if "code" in got else None
    if s.decode_body and body is None:
        strs, toks = marked("body")
        body = fm_index_generate(
            s.bart_model, fm_index, **encoder_batch(strs, toks),
            min_length=s.length, max_length=s.length, length_penalty=s.length_penalty, num_beams=s.beam,
            disable_fm_index=not constrained_generation, diverse_bs_groups=s.diverse_bs_groups,
            diverse_bs_penalty=s.diverse_bs_penalty, stop_at_count=s.stop_at_count, keep_history=True, topk=s.topk,
            logit_bias=bias, pending=True, **dec(s.bart_model))
    yield "body"
    if s.decode_titles and titles is None:
        strs, title_toks = marked("title")
        titles = fm_index_generate(
            s.bart_title_model, fm_index, **encoder_batch(strs, title_toks),
            min_length=1, max_length=TITLE_MAX_LENGTH, num_beams=s.beam, length_penalty=s.length_penalty,
            force_decoding_from=[s.title_bos_token_id], eos_token_id=s.title_eos_token_id,
            diverse_bs_groups=s.diverse_bs_groups, diverse_bs_penalty=s.diverse_bs_penalty, keep_history=True,
            disable_fm_index=not constrained_generation, topk=s.topk, logit_bias=bias, pending=True, **dec(s.bart_title_model))
    if s.decode_code and codes is None:           # retrieval.py:212-236
        strs, code_toks = marked("code")
        codes = fm_index_generate(
            s.bart_code_model, fm_index, **encoder_batch(strs, code_toks),
            min_length=1, max_length=TITLE_MAX_LENGTH, num_beams=s.beam, length_penalty=s.length_penalty,
            eos_token_id=s.code_eos_token_id, diverse_bs_groups=s.diverse_bs_groups, diverse_bs_penalty=s.diverse_bs_penalty,
            keep_history=True, force_decoding_from=[s.code_bos_token_id], disable_fm_index=not constrained_generation,
            logit_bias=bias, pending=True, **dec(s.bart_code_model))
    yield "decoding"

    # ---- segment 2: the hypotheses (waits for the decodes) ----
    # On the GPU the recorded history comes back as two arrays per decode and the filters below run on them (``_HypArrays``);
    # python lists -- what the reference's comprehensions work on -- are built for the survivors only.  Off the GPU (or with
    # settings the array form does not cover) the lists are built first, as ever.
    body_arr = title_arr = None
    use_arrays = (getattr(s, "array_filters", True) and s.force_decoding_second_token < 0 and not s.decode_code
                  and (body is None or body._packed is not None) and (titles is None or titles._packed is not None)
                  and (body is not None or titles is not None))
    if use_arrays:
        body_arr = _HypArrays(body.arrays()) if body is not None else None
        title_arr = _HypArrays(titles.arrays()) if titles is not None else None
        found_keys = [[] for _ in inputs]
        decoded = None
    else:
        found_keys = body.result() if body is not None else [[] for _ in inputs]
        decoded = titles.result() if titles is not None else None
    decoded_code = codes.result() if codes is not None else None
    yield "decoded"

    # ---- segment 3: filters, rescoring, query n-grams, unigram scores ----
    # The reference filters and rescores the body keys, then the query n-grams, then the title keys, each step ending in
    # a read-back.  None of the three depends on another, so here: the candidate lists of all three are built first,
    # ONE count launch filters them, the (up to) three rescorings are enqueued back to back, and only then are the
    # scores read back -- in the reference's order, into the same lists.
    if body_arr is not None:    # retrieval.py:85-90 on the arrays: empty keys dropped before each strip
        body_arr.drop_empty(); body_arr.strip_front(strip_ids)
        body_arr.drop_empty(); body_arr.strip_front(strip_ids)
        body_arr.drop_empty(); body_arr.strip_back(strip_ids)
        if s.min_length > 0:
            body_arr.require_length(s.min_length)
    elif s.decode_body:
        for fk in found_keys:   # retrieval.py:85-90
            fk[:] = [(sc, k[1:] if k[0] in strip_ids else k) for sc, k in fk if k]
            fk[:] = [(sc, k[1:] if k[0] in strip_ids else k) for sc, k in fk if k]
            fk[:] = [(sc, k[:-1] if k[-1] in strip_ids else k) for sc, k in fk if k]
            if s.min_length > 0:
                fk[:] = [(sc, k) for sc, k in fk if len(k) == s.min_length]
    cand = None
    if s.add_query_to_keys:
        if tokenised:
            # pre-tokenised queries (an extension): the word n-grams are token n-grams (seal_amd/query_keys.py)
            from .query_keys import token_ngram_keys
            cand = [token_ngram_keys(q, s) for q in base_tokens]
        else:
            if s.bart_tokenizer is None:
                raise RuntimeError("add_query_to_keys decomposes the query STRING into word n-grams (spaCy + tokenizer, "
                                   "reference retrieval.py:113-131): a tokenizer is needed for string queries")
            from .query_keys import query_ngram_keys
            cand = [query_ngram_keys(inp, s) for inp in inputs]
        cand = [[(0.0, k) for k in kk] for kk in cand]
    title_keys = None
    if title_arr is not None:   # retrieval.py:180-190 on the arrays
        title_arr.strip_back(strip_ids)
        if not s.partial_titles:
            title_arr.require_last(s.title_eos_token_id)
            if s.min_length > 0:
                title_arr.require_length(s.min_length + 1)
    elif s.decode_titles:
        title_keys = [[(sc, hyp) for sc, hyp in dec_] for dec_ in decoded]
        for fk in title_keys:   # retrieval.py:180-190
            if s.force_decoding_second_token >= 0:
                fk[:] = [(sc, k[:1] + k[2:]) for sc, k in fk if len(k) >= 3]
            fk[:] = [(sc, k[:-1] if k[-1] in strip_ids else k) for sc, k in fk]
            if not s.partial_titles:
                fk[:] = [(sc, k) for sc, k in fk if k[-1] == s.title_eos_token_id]
                if s.min_length > 0:
                    fk[:] = [(sc, k) for sc, k in fk if len(k) == (s.min_length + 1)]
            fk[:] = [(sc, [s.title_bos_token_id] + k if k[0] != s.title_bos_token_id else k) for sc, k in fk]
    code_keys = None
    if s.decode_code:
        code_keys = [[(sc, hyp) for sc, hyp in dec_] for dec_ in decoded_code]
        for fk in code_keys:    # retrieval.py:240-246
            if s.force_decoding_second_token >= 0:
                fk[:] = [(sc, k[:1] + k[2:]) for sc, k in fk if len(k) >= 2]
            fk[:] = [(sc, k[1:-1] if k[-1] in strip_ids else k[1:]) for sc, k in fk if k]
            if not s.partial_code:
                fk[:] = [(sc, k) for sc, k in fk if k and k[-1] == s.code_eos_token_id]
            fk[:] = [(sc, [s.code_bos_token_id] + k if k[0] != s.code_bos_token_id else k) for sc, k in fk if k]
    # retrieval.py:91, 130, 191, 247: get_count(k) > 0, one launch for all the lists
    n_q = len(inputs)
    if body_arr is not None or title_arr is not None:
        import numpy as np
        # body hypotheses, query n-grams, title hypotheses: one CSR, one launch.  A title key is counted with the title bos
        # in front where it does not start with it (retrieval.py:190 runs before the count of :191)
        parts_len, parts_tok, body_idx, title_idx = [], [], None, None
        if body_arr is not None:
            body_idx, ln, tk = body_arr.csr()
            parts_len.append(ln); parts_tok.append(tk)
        cand_flat = [k for kk in cand for _, k in kk if k] if cand is not None else []
        if cand_flat:
            parts_len.append(np.fromiter(map(len, cand_flat), dtype=np.int64, count=len(cand_flat)))
            parts_tok.append(np.fromiter((t for k in cand_flat for t in k), dtype=np.int64))
        title_extra = None
        if title_arr is not None:
            title_idx, ln, tk = title_arr.csr()
            first = title_arr._at(title_arr.start)[title_idx]
            title_extra = first != s.title_bos_token_id
            if title_extra.any():                  # rare: the bos in front of those keys, in place in the flat token stream
                offs = np.concatenate([[0], np.cumsum(ln)])[:-1]
                tk = np.insert(tk, offs[title_extra], s.title_bos_token_id)
                ln = ln + title_extra
            parts_len.append(ln); parts_tok.append(tk)
        lens = np.concatenate(parts_len) if parts_len else np.zeros(0, np.int64)
        counts = np.zeros(0, np.int64)
        if len(lens):
            offsets = np.zeros(len(lens) + 1, dtype=np.int64)
            np.cumsum(lens, out=offsets[1:])
            flat = np.concatenate(parts_tok)
            if hasattr(fm_index, "get_range_csr"):
                lo, hi = fm_index.get_range_csr(offsets, flat)
                counts = (hi - lo).astype(np.int64)
            else:               # an index without the CSR entry point (tests: the oracle behind the batched interface)
                counts = np.asarray(fm_index.get_count_batch([flat[offsets[i]:offsets[i + 1]].tolist() for i in range(len(lens))]), dtype=np.int64)
        a = 0
        if body_arr is not None:
            found_keys = body_arr.lists(body_idx[counts[a:a + len(body_idx)] > 0])
            a += len(body_idx)
        if cand is not None:
            it = iter(counts[a:a + len(cand_flat)].tolist())
            cand = [[k for _, k in kk if k and next(it) > 0] for kk in cand]
            a += len(cand_flat)
        if title_arr is not None:
            title_keys = title_arr.lists(title_idx[counts[a:a + len(title_idx)] > 0], prepend=s.title_bos_token_id)
    else:
        parts = [p for p in (found_keys if s.decode_body else None, cand, title_keys, code_keys) if p is not None]
        if parts:
            merged = _count_filter(fm_index, [fk for p in parts for fk in p])
            parts = [merged[j * n_q:(j + 1) * n_q] for j in range(len(parts))]
            it = iter(parts)
            if s.decode_body:
                found_keys = next(it)
            if cand is not None:
                cand = [[k for _, k in kk] for kk in next(it)]
            if title_keys is not None:
                title_keys = next(it)
            if code_keys is not None:
                code_keys = next(it)
    marked_rescoring = s.rescore and s.use_markers
    # the rescorings of the batch (retrieval.py:93-100, 139-149, 193-203, 249-263): the jobs of one model share one forward
    body_job = cand_job = title_job = code_job = None
    jobs, slots = [], []
    if s.decode_body and marked_rescoring:
        jobs.append((s.bart_model, base_tokens, found_keys, dict(
            length_penalty=0.0, strip_from_bos=bos_strip,
            strip_from_eos=[s.title_eos_token_id, s.code_eos_token_id, s.bart_model.config.eos_token_id], logit_bias=bias,
            encoded=getattr(body, "extra_encoded", None))))
        slots.append("body")
    if cand is not None:
        _, toks = marked("body")
        last_input_tokens = toks                # the reference re-binds `input_tokens` here (retrieval.py:139)
        # same encoder input as the body decode (' || body || +'): its encoder states are reused where there are any
        reuse = (body.enc, body.attention_mask) if (tokenised and body is not None and body.enc is not None) else None
        jobs.append((s.bart_model, toks, cand, dict(length_penalty=0.0, logit_bias=bias, encoded=reuse)))
        slots.append("cand")
    if title_keys is not None and marked_rescoring:
        jobs.append((s.bart_title_model, title_toks, title_keys, dict(
            length_penalty=0.0, strip_from_bos=bos_strip, strip_from_eos=[s.bart_model.config.eos_token_id], logit_bias=bias,
            encoded=(titles.enc, titles.attention_mask) if tokenised and titles.enc is not None else None)))
        slots.append("title")
    if code_keys is not None and marked_rescoring:     # retrieval.py:249-263
        jobs.append((s.bart_code_model, code_toks, code_keys, dict(
            length_penalty=0.0, strip_from_bos=bos_strip, strip_from_eos=[s.bart_model.config.eos_token_id], logit_bias=bias,
            encoded=(codes.enc, codes.attention_mask) if tokenised and codes.enc is not None else None)))
        slots.append("code")
    yield "filtered"                             # the candidate lists are final; what follows launches GEMMs
    if jobs:
        gate = s.__dict__.get("_gemm_gate")
        if gate is not None:
            gate()              # the overlapped search: the rescoring forward (library GEMMs) waits for the decodes enqueued so far
        pend = dict(zip(slots, rk.rescore_keys_multi(jobs, pending=True)))
        body_job, cand_job, title_job, code_job = pend.get("body"), pend.get("cand"), pend.get("title"), pend.get("code")
    yield "rescoring"                            # everything of this batch up to the scores is enqueued; nothing read back yet
    if body_job is not None:
        found_keys = body_job.result()
    if cand_job is not None:
        for fk, nfk in zip(found_keys, cand_job.result()):
            fk += nfk
    if title_keys is not None:
        if title_job is not None:
            title_keys = title_job.result()
        for nfk, fk in zip(title_keys, found_keys):
            fk += nfk
    if code_keys is not None:
        if code_job is not None:
            code_keys = code_job.result()
        for nfk, fk in zip(code_keys, found_keys):
            fk += nfk

    def gemm_gate():
        # (overlapped search) a forward AFTER the "rescoring" yield -- the un-marked rescoring, a separate scorer model's unigram scores, a
        # forced second token -- is a library-GEMM phase too: it waits for the decodes released meanwhile, and the caller fences behind it
        g = s.__dict__.get("_gemm_gate")
        if g is not None:
            g()
    if s.rescore and not s.use_markers:
        gemm_gate()
        found_keys = rk.rescore_keys(s.bart_scorer_model, last_input_tokens, found_keys, batch_size=100, length_penalty=0.0,
                                     strip_from_bos=bos_strip, strip_from_eos=[s.bart_model.config.eos_token_id])

    found_keys = [rk.deduplicate(fk) for fk in found_keys]
    found_keys = [[(n, sc) for sc, n in fk] for fk in found_keys]   # flip to (ngram, score), retrieval.py:284

    if (s.unigram_scores and body is not None and body.first_logits is not None and s.bart_scorer_model is s.bart_model
            and s.force_decoding_second_token < 0):
        # compute_unigram_scores (keys.py:145-176) = the model's next-token distribution after the decoder start token
        # for the ' || body' input: exactly the first step of the body decode above
        unigram = torch.log_softmax(body.first_logits.float(), dim=-1).double().cpu().numpy()
        return list(zip(found_keys, unigram))
    if s.unigram_scores:
        _, toks = marked("body")
        gemm_gate()
        unigram = rk.compute_unigram_scores(
            s.bart_scorer_model, toks, fm_index,
            prefix=[s.force_decoding_second_token] if s.force_decoding_second_token >= 0 else [], logit_bias=bias,
            tolist=False)
        # one D2H copy; float64 views of the fp32 log-probs == what .tolist() would hold
        unigram = unigram.double().cpu().numpy()
        return list(zip(found_keys, unigram))
    return found_keys


def _low_priority_stream(dev):
    """A stream of the LOWEST priority for the searcher's rescoring phase (filters, rescoring forward, unigram scores).  Not for the priority
    as such: HIP maps streams onto a few hardware queues per priority level, and at the default priority the rescoring stream has -- in four
    runs of seven -- landed on the hardware queue of the caller's stream, where the decodes of the next two batches are already queued: the
    first rescoring of a call then ran ~120 ms late, behind them (363 instead of 405 queries/s over 20 batches; bimodal from run to run).  A stream
    of another priority level gets a queue of that level: the index's service stream took the high one for the same reason in round 5 (index.py),
    the rescoring -- the phase nothing on the GPU waits for -- takes the low one (8 of 8 runs at 399-412 queries/s).  torch.cuda.Stream offers
    only the levels -1 and 0 (HIP: -1 .. 1), so the stream is made by the runtime and wrapped."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        raw = ctypes.c_void_p()
        with torch.cuda.device(dev):
            lo, hi = ctypes.c_int(), ctypes.c_int()
            if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0 and lo.value > 0 \
                    and hip.hipStreamCreateWithPriority(ctypes.byref(raw), ctypes.c_uint(1), ctypes.c_int(lo.value)) == 0 and raw.value:   # 1 = hipStreamNonBlocking
                return torch.cuda.ExternalStream(raw.value, device=dev)
    except (OSError, AttributeError):
        pass
    return torch.cuda.Stream(device=dev)


class SEALSearcher:
    """Drop-in for ``seal.retrieval.SEALSearcher`` (reference retrieval.py:399-811)."""

    DEFAULTS = {
        "backbone": "facebook/bart-large",
        "fairseq_checkpoint": True,
        "length": 10,
        "min_length": 0,
        "length_penalty": 0.0,
        "scoring_length_penalty": 0.0,
        "repetition_penalty": 0.8,
        "score_exponent": 2.0,
        "beam": 15,
        "max_hits": 1500,
        "fully_score": 1500,
        "skip_frequent_keys": 10_000_000,
        "add_query_to_keys": True,
        "batch_size": 20,
        "jobs": 1,
        "progress": False,
        "free_generation": False,
        "use_fm_index_frequency": True,
        "unigram_scores": True,
        "add_best_unigrams_to_ngrams": True,
        "use_top_k_ngrams": 5000,
        "sort_by_length": False,
        "sort_by_freq": False,
        "print_n_doc": False,
        "allow_overlaps": False,
        "diverse_bs_groups": 1,
        "diverse_bs_penalty": 0.0,
        "rescore": True,
        "detokenize": True,
        "include_keys": False,
        "single_key": 0.0,
        "unigrams_ignore_free_places": False,
        "use_markers": True,
        "value_conditioning": True,
        "decode_body": True,
        "decode_titles": True,
        "decode_code": False,
        "partial_code": False,
        "partial_titles": False,
        "smoothing": 5.0,
        "stop_at_count": 0,
        "topk": 0,
        "force_decoding_second_token": -1,
    }

    def __init__(self, fm_index: FMIndex, bart_tokenizer, bart_model, bart_scorer_model=None, bart_title_model=None,
                 bart_code_model=None, **params):
        self.fm_index = fm_index
        self.docid2idx = {k: i for i, k in enumerate(self.fm_index.labels)} if self.fm_index.labels else {}
        self.bart_tokenizer = bart_tokenizer
        self.bart_model = bart_model
        self.bart_scorer_model = bart_scorer_model if bart_scorer_model is not None else bart_model
        self.bart_title_model = bart_title_model if bart_title_model is not None else bart_model
        self.bart_code_model = bart_code_model if bart_code_model is not None else bart_model
        self.num_docs = fm_index.n_docs
        self.docids = fm_index.labels
        self.set_params(params)
        # extension: ids of the marker suffixes for pre-tokenised queries
        self.marker_token_ids: Dict[str, List[int]] = params.get(
            "marker_token_ids", {"body": [45056, 809], "title": [45056, 1270], "code": [45056, 3260], "+": [45056, 2055]})
        # extension: stop aggregate_evidence after the first stage (keys.py:311-364)
        self.first_stage_only: bool = params.get("first_stage_only", False)
        # extension: False = evidence aggregation through the host checker routines instead of the GPU kernels
        self.gpu_aggregate: bool = params.get("gpu_aggregate", True)
        # extension: the post-filters of the decodes run on the history as arrays (GPU decodes); False = on python lists
        self.array_filters: bool = bool(params.get("array_filters", True))
        # extension: the decodes of a batch that share a model (body, title[, code]) run as ONE loop, rows stacked
        self.joint_decode: bool = bool(params.get("joint_decode", True))
        # extension: enqueue the next batch's decodes before this batch's rescoring / aggregation (same thread, second stream)
        self.overlap: bool = bool(params.get("overlap", True))
        self.overlap_depth: int = int(params.get("overlap_depth", 2))     # batches of decodes kept enqueued ahead
        if params.get("pipeline"):
            # round 3's `pipeline=N` (N batches in flight on worker threads, one stream each) is gone: several GEMM streams at once are what
            # stalled the GPU; it must not be ignored silently
            raise ValueError("SEALSearcher(pipeline=N) was removed in round 4 (several library-GEMM streams at once can stall the GPU): "
                             "use overlap=True / overlap_depth (the default) instead")
        # the decode and the rescoring phase (the two that run library GEMMs) ALTERNATE on the GPU instead of sharing it: two stream-K
        # GEMM streams in flight at once stalled the GPU for ever.  False restores round 3's behaviour.
        self.exclusive_gemm_streams: bool = bool(params.get("exclusive_gemm_streams", True))
        self.i_know_two_gemm_streams_can_stall: bool = bool(params.get("i_know_two_gemm_streams_can_stall", False))
        # ... where "the decode" means the part of it that runs library GEMMs: its model steps do not (round 6: every product of a step is the
        # hand-written kernel), so a batch's rescoring runs BESIDE the next batch's decode steps.  False: rescoring after the whole decode (round 5).
        self.rescore_beside_decode: bool = bool(params.get("rescore_beside_decode", True))
        # extension (synthetic benchmarks): per-query additive bias on the model's next-token logits, [batch, vocab]
        self.logit_bias = None
        if "bart" in self.backbone:   # retrieval.py:480-491
            self.title_bos_token, self.title_bos_token_id = "</s>", 2
            self.title_eos_token, self.title_eos_token_id = "@@", 49314
            self.code_bos_token, self.code_bos_token_id = "@@", 49314
            self.code_eos_token, self.code_eos_token_id = "||", 45056
            self.prepend_space = True
            self.strip_token_ids = (0, 2)
        elif "t5" in self.backbone:
            self.title_bos_token, self.title_bos_token_id = "</s>", 1
            self.title_eos_token, self.title_eos_token_id = "<extra_id_99>", 32000
            self.code_bos_token, self.code_bos_token_id = "<extra_id_99>", 32000
            self.code_eos_token, self.code_eos_token_id = "<extra_id_98>", 32001
            self.prepend_space = False
            self.strip_token_ids = (0, 1)
        else:
            raise NotImplementedError
        for k in ("title_eos_token_id", "title_bos_token_id", "code_eos_token_id", "code_bos_token_id"):
            if k in params:   # extension: small-vocabulary test models
                setattr(self, k, params[k])

    @property
    def device(self):
        return next(self.bart_model.parameters()).device

    @device.setter
    def device(self, device: str):
        self.bart_model.to(device)

    def set_params(self, params):
        for key, val in self.DEFAULTS.items():
            setattr(self, key, params.get(key, val))

    @classmethod
    def add_args(cls, parser):
        parser.add_argument("--fm_index", required=True, type=str)
        parser.add_argument("--checkpoint", required=False, type=str)
        parser.add_argument("--checkpoint_scorer", required=False, type=str, default=None)
        parser.add_argument("--checkpoint_title", required=False, type=str, default=None)
        parser.add_argument("--checkpoint_code", required=False, type=str, default=None)
        parser.add_argument("--device", default="cuda:0", type=str)
        for name, value in cls.DEFAULTS.items():
            if value is True:
                parser.add_argument(f"--dont_{name}", action="store_false", dest=name)
            elif value is False:
                parser.add_argument(f"--{name}", action="store_true")
            else:
                parser.add_argument(f"--{name}", required=False, type=type(value), default=value)

    @classmethod
    def from_args(cls, args):
        params = {name: getattr(args, name) for name in cls.DEFAULTS}
        return cls.load(args.fm_index, args.checkpoint, bart_scorer_model_path=args.checkpoint_scorer,
                        bart_title_model_path=args.checkpoint_title, bart_code_model_path=args.checkpoint_code,
                        device=args.device, **params)

    @staticmethod
    def load_fm_index(fm_index_path: str):
        logger.warning(f"initializing FM-index from {fm_index_path}")
        index = FMIndex.load(fm_index_path)
        logger.warning(f"FM-index resident in HBM ({index.device_bytes() // 1024 ** 2} MBs)")
        return index

    @staticmethod
    def load_bart(bart_model_path: str, device: str = "cuda:0", backbone="facebook/bart-large", fairseq_checkpoint=True):
        """reference retrieval.py:561-592 (needs the HF hub files of ``backbone``)."""
        from transformers import AutoConfig, AutoModelForSeq2SeqLM, AutoTokenizer
        config = AutoConfig.from_pretrained(backbone)
        config.forced_bos_token_id = None
        tokenizer = AutoTokenizer.from_pretrained(backbone)
        if bart_model_path:
            from .utils import load_state_dict_from_fairseq_checkpoint, load_state_dict_from_lightning_checkpoint
            model = AutoModelForSeq2SeqLM.from_config(config)
            model.resize_token_embeddings(len(tokenizer))
            if fairseq_checkpoint:
                load_state_dict_from_fairseq_checkpoint(model, bart_model_path)
            else:
                load_state_dict_from_lightning_checkpoint(model, bart_model_path)
        else:
            model = AutoModelForSeq2SeqLM.from_pretrained(backbone)
            model.resize_token_embeddings(len(tokenizer))
        model.config.forced_bos_token_id = None
        model.eval()
        if hasattr(model, "final_logits_bias"):   # retrieval.py:583-588
            model.config.add_bias_logits = True
            model.final_logits_bias[0, tokenizer.pad_token_id] = float("-inf")
            model.final_logits_bias[0, tokenizer.bos_token_id] = float("-inf")
            model.final_logits_bias[0, tokenizer.mask_token_id] = float("-inf")
        model.to(device)
        return tokenizer, model

    @classmethod
    def load(cls, fm_index_path, bart_model_path, device="cuda:0", **params):
        fm_index = cls.load_fm_index(fm_index_path)
        kw = dict(backbone=params.get("backbone", "facebook/bart-large"), fairseq_checkpoint=params.get("fairseq_checkpoint", True))
        bart_tokenizer, bart_model = cls.load_bart(bart_model_path, device, **kw)
        extra = {}
        for name in ("scorer", "title", "code"):
            path = params.get(f"bart_{name}_model_path")
            extra[f"bart_{name}_model"] = cls.load_bart(path, device, **kw)[1] if path is not None else None
        return cls(fm_index, bart_tokenizer, bart_model, **extra, **params)

    def search(self, query, k: int = 10, added_documents=None, detokenize=True) -> List[SEALDocument]:
        if added_documents is not None:
            added_documents = [added_documents]
        return self.batch_search([query], k=k, added_documents=added_documents, detokenize=True)[0]

    def batch_search(self, queries, k: int = 10, added_documents=None, detokenize=None) -> List[List[SEALDocument]]:
        """reference retrieval.py:649-691"""
        if detokenize is None:
            detokenize = self.detokenize
        if self._overlapped() and added_documents is None:
            # the next batch's decodes are enqueued before this batch's rescoring / aggregation start (second stream)
            ranked = self._overlapped_results(queries, keep=k)
        else:
            keys = self.batch_generate_keys(queries)
            if added_documents is not None:
                if self.unigram_scores:
                    keys = ((kk, us, added_documents[i]) for i, (kk, us) in enumerate(keys))
                else:
                    keys = ((kk, None, added_documents[i]) for i, kk in enumerate(keys))
            ranked = self.batch_retrieve_from_keys(keys, keep=k)
        key_info = {}
        retrieved = []
        # streamed: a query's (up to fully_score) ranked documents are cut to k as soon as they arrive
        for query, (res, _) in zip(queries, ranked):
            docs = []
            if hasattr(res, "top") and not self.include_keys:
                # the GPU aggregation's records: ids, scores and document tokens of the top k straight from the fetched arrays
                ids, scores, toks = res.top(k)
                for idx, score, full in zip(ids, scores, toks):
                    doc = SEALDocument(idx, score, self.fm_index, self.bart_tokenizer, delim1=self.title_eos_token_id,
                                       delim2=self.code_eos_token_id, keys=None, query=query)
                    doc._raw_tokens = full            # list-like over the fetched tokens (index / slices / == as a list); a list on first use
                    docs.append(doc)
                retrieved.append(docs)
                continue
            for idx, info in islice(res.items(), k):
                score, kk, full = info[0], info[1], (info[3] if len(info) == 5 else None)
                doc = SEALDocument(idx, score, self.fm_index, self.bart_tokenizer, delim1=self.title_eos_token_id,
                                   delim2=self.code_eos_token_id, keys=None, query=query)
                if self.include_keys:
                    for key, _ in kk:
                        key = tuple(key)
                        if key not in key_info:
                            text = (self.bart_tokenizer.decode(list(key), clean_up_tokenization_spaces=False)
                                    if self.bart_tokenizer is not None else None)
                            key_info[key] = (text, self.fm_index.get_count(list(key)))
                    doc.keys = [(*key_info[tuple(key)], sc) for key, sc in kk]
                doc._raw_tokens = list(full) if full is not None else None     # a real list: split_tokens uses .index()
                docs.append(doc)
            retrieved.append(docs)
        if detokenize and self.bart_tokenizer is not None:
            return self.detokenize_retrieved(retrieved)
        return retrieved

    # ------------------------------------------------------------------
    # the next batches' decodes enqueued ahead of a batch's rescoring / aggregation
    # ------------------------------------------------------------------
    def _overlapped(self) -> bool:
        return bool(self.overlap) and self.device.type == "cuda" and hasattr(self.fm_index, "handle")

    def _overlapped_results(self, queries, keep=None):
        """``(results, all_ngrams)`` per query, in query order, one batch after the other as always -- but the decodes of
        batch i+1 are enqueued (on the caller's stream) as soon as the hypotheses of batch i are on the host, and batch
        i's filters, rescoring, unigram scores and evidence aggregation then run on a second stream: the GPU works on
        the next decode while the host walks through this batch's python, and the rescoring GEMMs fill the gaps.  One
        thread, one index handle (a decode uses the constraint workspace, the rest of a batch does not), no result
        depends on the schedule."""
        import os
        import torch
        dev = self.device
        post = self.__dict__.get("_post_stream")
        if post is None:
            post = self.__dict__["_post_stream"] = _low_priority_stream(dev)
        params = self._aggregate_params()
        constrained = not self.free_generation
        batches, offsets, off = [], [], 0
        for b in _chunks(queries, self.batch_size):
            batches.append(b)
            offsets.append(off)
            off += len(b)
        if not batches:
            return
        post.wait_stream(torch.cuda.current_stream(dev))      # whatever the caller queued (weights, logit bias) is visible
        import os, sys, time
        tm = os.environ.get("SEAL_OVERLAP_TIMING")
        # decodes are independent of everything else and of each other: keep `overlap_depth` batches' worth of them
        # enqueued AHEAD of the batch whose post-processing (filters, rescoring, aggregation -- phases that end in a
        # device -> host read-back, i.e. pipeline bubbles) is running, so that the GPU never drains.  They share the
        # decoder's static buffers and the index workspace in stream order; each keeps its own history tensors.
        # ONE library GEMM stream at a time (round 4).  The decodes (caller's stream) and the rescoring forward (post stream) both run
        # hipBLASLt GEMMs, all stream-K kernels whose workgroups wait for partner workgroups through flags; two of them in flight at
        # once on one GPU stalled for ever -- reproducibly with some algorithm picks (round 3's TunableOp pick for lm_head, every fp16
        # pick of the split GEMM: profiles/r4_hang_*.txt, r4_split_gemm_two_gemm_streams_stalls.txt), never with launches serialised.
        # So the two GEMM-bearing phases ALTERNATE on the GPU, enforced with events: a batch's rescoring waits for the decodes enqueued
        # before it, the next decode waits for that rescoring; the aggregation (index kernels, rocPRIM sorts: no library GEMM) still
        # overlaps both on the index's own stream.  The host keeps one more batch of decodes enqueued ahead (depth 2), so that the GPU
        # has the next decode queued while the host waits for a batch's scores: throughput is bound by the host loop or by
        # decode + rescoring, whichever is longer, as before.
        exclusive = bool(getattr(self, "exclusive_gemm_streams", True))
        if not exclusive:
            # two library-GEMM streams at once is the configuration that stalled the GPU for ever: it is for
            # reproducing that stall (tools/soak.py), never something a caller gets by flipping one switch
            if not getattr(self, "i_know_two_gemm_streams_can_stall", False):
                raise RuntimeError("exclusive_gemm_streams=False lets the decode and the rescoring forward run hipBLASLt stream-K GEMMs on two "
                                   "streams at once, which has stalled the GPU for ever (DESIGN.md section 9); pass "
                                   "i_know_two_gemm_streams_can_stall=True as well to do it anyway")
            if not self.__dict__.get("_warned_two_gemm_streams"):
                self.__dict__["_warned_two_gemm_streams"] = True
                print("[seal_amd] WARNING: decode and rescoring share the GPU (two library-GEMM streams): this configuration can stall",
                      file=sys.stderr, flush=True)
        depth = max(1, int(self.overlap_depth if exclusive else 1))
        # What the host does between "the scores of batch i are back" and "the rescoring of batch i+1 is enqueued" decides whether the GPU
        # runs dry after decode(i+2) (see the loop): interleaved -- the aggregation's launches, the rescoring enqueue, the aggregation's
        # read-back.  (The whole aggregation first, round 4's order, and the rescoring first lost on one box: profiles/r5_rescore_ahead_ab.txt.)
        ahead = []                                            # generators whose decodes are enqueued, oldest first
        nxt_i = 0
        main = torch.cuda.current_stream(dev)
        fence = {"decode": None, "rescore": None}             # events behind the newest enqueued decode / rescoring

        def after(kind, stream):
            if exclusive:
                ev = torch.cuda.Event()
                ev.record(stream)
                fence[kind] = ev

        def wait_for(kind, stream):
            if exclusive and fence[kind] is not None:
                stream.wait_event(fence[kind])
        marks = []                                            # SEAL_OVERLAP_TIMING=2: (label, event) at the phase boundaries, in GPU time

        def mark(label, stream):
            if tm == "2":
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(stream)
                marks.append((label, ev))

        # Round 6: the decode's model STEPS hold no library GEMM (every product of a step runs in sealnn_hgemm_nt, which has no inter-workgroup
        # hand-off: split_gemm.HAND_CONFIGS); what does is the decode's PREFIX -- encoder, cross-attention K / V, the shared first step.  So the
        # fence a rescoring waits for is recorded right behind that prefix (the step decoder calls back once its first step is enqueued and
        # everything after it is library-free), not behind the whole decode: GPU order  prefix(i+1) -> [steps(i+1)  ||  rescoring(i)] ->
        # prefix(i+2) [waits for rescoring(i)] -> ...  -- still at most ONE stream of stream-K kernels at any time, and no phase boundary at
        # which the GPU drains: 380 -> 398 queries/s (profiles/r6_overlap_timeline.txt).  The two phases SHARE the chip rather than hide in each
        # other: beside the rescoring's chip-filling GEMMs a decode takes 48 ms instead of 38.5 and the rescoring 20 instead of 10.9 (a decode
        # kernel's few hundred workgroups wait for compute units the rescoring's hold); a high-priority decode stream changed nothing.
        # A decoder whose steps are not library-free (bf16 storage, split_gemm.HAND_GEMM off, other geometries) never calls back: the fence then sits
        # behind the whole decode, as in round 5.
        overlap_steps = exclusive and bool(getattr(self, "rescore_beside_decode", True))

        class _PrefixFence:
            """while active, the step decoders of this searcher's models record the "decode" fence behind their library-GEMM prefix"""
            def __init__(self_f):
                self_f.called = False
                self_f.models = [m for m in {id(m): m for m in (self.bart_model, getattr(self, "bart_title_model", None),
                                                                  getattr(self, "bart_code_model", None)) if m is not None}.values()]

            def __enter__(self_f):
                if overlap_steps:
                    def done():
                        self_f.called = True
                        after("decode", main)
                        mark("decode prefix ends", main)
                    for m in self_f.models:
                        m._seal_after_library_prefix = done
                return self_f

            def __exit__(self_f, *exc):
                for m in self_f.models:
                    m.__dict__.pop("_seal_after_library_prefix", None)
                return False

        def enqueue_next(upto="decoding"):
            """starts the next batch: its body decode is enqueued (``upto="body"``), or both decodes"""
            nonlocal nxt_i
            g = _batch_steps(self, batches[nxt_i], constrained, offsets[nxt_i])
            wait_for("rescore", main)
            mark("decode %d begins" % nxt_i, main)
            with _PrefixFence() as pf:
                state = next(g)                               # "body": the body decode is enqueued behind the earlier ones
                if upto == "decoding":
                    state = next(g)
            if not pf.called:
                after("decode", main)
            mark("decode %d ends" % nxt_i, main)
            ahead.append([g, state])
            nxt_i += 1

        ORDER = {"body": 0, "decoding": 1, "decoded": 2, "filtered": 3, "rescoring": 4}

        def advance(entry, upto):
            if ORDER[entry[1]] >= ORDER[upto]:
                return                                        # (already there: a batch's filters / rescoring may have been run one iteration early)
            decodes = entry[1] == "body"                      # a title decode still to be enqueued
            if decodes:
                wait_for("rescore", main)
            while entry[1] != upto:
                if decodes and entry[1] == "body":
                    with _PrefixFence() as pf:
                        entry[1] = next(entry[0])
                    if entry[1] == "decoding" and not pf.called:
                        after("decode", main)
                    continue
                entry[1] = next(entry[0])

        def to_rescoring(entry):
            """the batch's hypotheses on the host, its filters run, its rescoring forward enqueued on the post stream (fenced behind the decodes
            enqueued so far); entry[2] = how often its gate was called"""
            if ORDER[entry[1]] >= ORDER["rescoring"]:
                return
            advance(entry, "decoded")
            gated = [0]
            entry.append(gated)

            def gate():
                gated[0] += 1
                wait_for("decode", post)
                mark("rescoring begins", post)
            entry.append(gate)
            self.__dict__["_gemm_gate"] = gate                # (read by _batch_steps when it reaches a forward)
            try:
                with torch.cuda.stream(post):
                    advance(entry, "rescoring")
            finally:
                self.__dict__.pop("_gemm_gate", None)
            after("rescore", post)
            mark("rescoring ends", post)

        def to_filtered(entry):
            """the batch's hypotheses on the host and its post-filters run (one count launch and its read-back, on the aggregation's stream:
            the post stream may be busy with the previous batch's rescoring) -- everything of "rescoring" that needs no GEMM"""
            advance(entry, "decoded")
            with torch.cuda.stream(agg_stream):
                advance(entry, "filtered")
        # the aggregation's own launches (key counts, locate, evidence kernels) and the host's waits for them: the index's retrieval stream,
        # NOT the post stream -- a count read-back there would queue behind the next batch's rescoring forward (round 5)
        # (high priority, as the index's own stream: index.py _side_stream)
        prio = -1
        agg_stream = self.__dict__.get("_agg_stream")
        if agg_stream is None or agg_stream.device != dev or self.__dict__.get("_agg_stream_priority") != prio:
            agg_stream = self.__dict__["_agg_stream"] = torch.cuda.Stream(device=dev, priority=prio)
            self.__dict__["_agg_stream_priority"] = prio
        enqueue_next()
        if exclusive and len(batches) > 1 and depth > 1:
            enqueue_next()                                    # one decode ahead of the batch the loop starts with
        held = None                                           # results of the batch before the current one, not handed out yet
        for i in range(len(batches)):
            t0 = time.perf_counter()
            cur = ahead.pop(0)
            advance(cur, "decoded")                           # (title decode enqueued if it was not;) hypotheses of batch i on the host
            t1 = time.perf_counter()
            # The next batch's decodes go in two halves around this batch's rescoring: its body decode now, its title decode
            # once this batch's filters and rescorings are enqueued -- so the GPU still has decode work queued while the host
            # reads the scores back and walks through the aggregation (phases that end in a read-back).
            if not exclusive:
                while nxt_i < len(batches) and len(ahead) < depth:
                    enqueue_next("body" if len(ahead) == 0 else "decoding")
            t2 = time.perf_counter()
            # (the filters' count launch and copies run beside the decodes; only the rescoring forward is fenced, from inside the step)
            to_rescoring(cur)                                 # (a no-op when the previous iteration did it already, below)
            if exclusive:
                # GPU order: decode(i+1) [enqueued one iteration ago] -> rescoring(i) [it waits for that decode only] ->
                # decode(i+2) [enqueued now: it waits for the rescoring] -- the scores of batch i are back after ONE decode and the
                # GPU has the next decode queued while the host aggregates batch i
                for entry in ahead:
                    advance(entry, "decoding")
                while nxt_i < len(batches) and len(ahead) < depth:
                    enqueue_next("decoding")
            elif ahead:
                advance(ahead[0], "decoding")                 # the next batch's title decode, on the caller's stream
            # this batch's rescorings are enqueued and the host is about to wait for their scores: the time to hand the
            # PREVIOUS batch's results to the caller (who builds documents from them): its python runs under that wait
            if held is not None:
                yield from held
                held = None
            interleave = exclusive and bool(ahead)
            if interleave:
                # the host is about to wait for this batch's scores (the GPU is in its rescoring): the time for the NEXT batch's filters -- its
                # decode finished before this rescoring began, so its hypotheses are there; nothing of it needs the busy streams
                to_filtered(ahead[0])
                if nxt_i >= len(batches):
                    # no decode left to enqueue: nothing on the GPU but this batch's rescoring, and the LAST batch's should not wait for the host to walk
                    # through this batch's scores and aggregation plan first (the scores come back through their own event: keys._PendingRescore).
                    # Only at the end of a call: in its middle the same order costs a fifth of the throughput (452 .. 461 -> 358 .. 366 queries/s: the
                    # rescoring then runs beside the first, widest steps of the next decode instead of its later ones)
                    to_rescoring(ahead[0])
            with torch.cuda.stream(post):
                gated, gate = cur[2], cur[3]
                late = gated[0]
                self.__dict__["_gemm_gate"] = gate            # (forwards after the "rescoring" yield: non-default configurations)
                try:
                    next(cur[0])
                    raise RuntimeError("_batch_steps yielded more often than expected")
                except StopIteration as done:
                    keys = done.value
                finally:
                    self.__dict__.pop("_gemm_gate", None)
                if gated[0] != late:
                    # a forward ran after the yield (non-default configurations): the decodes enqueued from here on wait for it as well
                    after("rescore", post)
            t3 = time.perf_counter()
            mark("  host: scores of batch %d are back" % i, agg_stream)      # (an idle stream: the event's GPU time is the host's "now")
            # The scores of batch i are back: the GPU is in decode(i+2) (~38 ms), and rescoring(i+1) has to be in its queue before that decode
            # ends or the GPU idles for the difference.  The host has two things to do: the aggregation of batch i (~11 ms: key lists, one
            # count launch, key scoring in C++, fmi_dev_aggregate = 3.5 ms of index kernels, two read-backs) and the rescoring enqueue of batch
            # i+1 (~16 ms: filters, prefix tree, graph replay).  Round 4's order -- the whole aggregation, then (next iteration) filters and
            # rescoring -- has the rescoring in the queue ~33 ms into the decode: in time on a fast host, late on a slower one (a box at
            # 294 queries/s where others gave 337; 299 vs 318 with the collector walking earlier results).  Rescoring first is never late, but
            # then the index kernels land beside the rescoring's chip-filling GEMMs instead of beside the decode's small ones, which leave
            # CUs idle: 326 vs 340 queries/s on a fast host (profiles/r5_rescore_ahead_ab.txt).  So the two are interleaved: the aggregation's
            # host half and its launches first (kernels beside the decode), the rescoring enqueue while they run, then the read-back; and
            # the filters of batch i+1 (no GEMM) have moved in front of the wait for the scores above.
            jobs = [(kk[0], kk[1]) if isinstance(kk, tuple) else (kk, None) for kk in keys]
            if tm == "2":
                self.fm_index.__dict__["_agg_mark"] = mark
            try:
                with torch.cuda.stream(agg_stream):
                    fetch = rk.aggregate_evidence_batch(jobs, self.fm_index, two_phase=interleave, keep=keep, gpu_aggregate=self.gpu_aggregate,
                                                        want_ngrams=False, **params)
            finally:
                self.fm_index.__dict__.pop("_agg_mark", None)      # (only this call's _run_plan may record into this loop's list)
            t4 = time.perf_counter()
            mark("  host: aggregation of batch %d launched" % i, agg_stream)
            if tm == "2" and hasattr(self.fm_index, "_side_stream"):
                mark("  index stream: aggregation kernels of batch %d done" % i, self.fm_index._side_stream(dev))
            if interleave:
                try:
                    to_rescoring(ahead[0])
                except BaseException:
                    try:
                        fetch()                               # (the plan and the index's aggregation buffers belong to the pending call)
                    except Exception:
                        pass
                    raise
                mark("  host: next rescoring enqueued", agg_stream)
                t5 = time.perf_counter()
                with torch.cuda.stream(agg_stream):
                    out = fetch()
            else:
                t5, out = t4, fetch
            agg_stream.synchronize()
            mark("  host: aggregation of batch %d fetched" % i, agg_stream)
            if tm:
                print("[overlap] batch %d: waited %.1f ms for its decodes; enqueued further decodes in %.1f ms; filters / rescoring / next filters / scores %.1f ms; "
                      "aggregation %.1f ms + %.1f ms around the next batch's rescoring enqueue %.1f ms" %
                      (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (time.perf_counter() - t5) * 1e3, (t5 - t4) * 1e3), file=sys.stderr, flush=True)
            held = out
        if marks:
            torch.cuda.synchronize(dev)
            t_first, prev = marks[0][1], None
            for label, ev in sorted(marks, key=lambda m: t_first.elapsed_time(m[1])):
                at = t_first.elapsed_time(ev)
                print("[overlap] GPU time %9.2f ms (+%6.2f)  %s" % (at, at - (prev if prev is not None else at), label), file=sys.stderr, flush=True)
                prev = at
        if held is not None:
            yield from held

    def detokenize_retrieved(self, retrieved):
        """reference retrieval.py:693-712"""
        for docs in retrieved:
            for d in docs:
                title, body = d.split_tokens(d.raw_tokens())
                d._title, d._body = self._batch_detokenize([title, body])
        return retrieved

    def _batch_detokenize(self, seqs):
        return [self.bart_tokenizer.decode(seq, skip_special_tokens=True, clean_up_tokenization_spaces=False).strip()
                if seq else "" for seq in seqs]

    def generate_keys(self, query):
        return next(self.batch_generate_keys([query]))

    def batch_generate_keys(self, queries):
        return batch_generate_keys(self, queries, constrained_generation=not self.free_generation)

    def _host_pool(self):
        """``jobs`` worker PROCESSES for the host-side first stage (the reference forks ``jobs``
        processes over queries, retrieval.py:762-775; here only the GPU-free bookkeeping is shipped,
        the index never leaves the GPU).  Spawned once, reused across calls."""
        pool = self.__dict__.get("_pool")
        if pool is None:
            import multiprocessing
            import os
            from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
            pool = ProcessPoolExecutor(max_workers=int(self.jobs), mp_context=multiprocessing.get_context("spawn"))
            self.__dict__["_pool"] = pool
        return pool

    def retrieve_from_keys(self, keys, defer=None, keep=None):
        """reference retrieval.py:720-754"""
        unigram_scores = None
        if isinstance(keys, tuple) and len(keys) == 1:
            keys = keys[0]
        elif isinstance(keys, tuple) and len(keys) == 2:
            keys, unigram_scores = keys
        elif isinstance(keys, tuple) and len(keys) == 3:
            keys, unigram_scores, _ = keys
        return rk.aggregate_evidence(
            ngrams_and_scores=keys, unigram_scores=unigram_scores, index=self.fm_index,
            max_occurrences_1=self.max_hits, n_docs_complete_score=self.fully_score, alpha=self.score_exponent,
            beta=self.repetition_penalty, length_penalty=self.scoring_length_penalty,
            use_fm_index_frequency=self.use_fm_index_frequency,
            add_best_unigrams_to_ngrams=self.add_best_unigrams_to_ngrams, use_top_k_unigrams=self.use_top_k_ngrams,
            sort_by_length=self.sort_by_length, sort_by_freq=self.sort_by_freq, smoothing=self.smoothing,
            allow_overlaps=self.allow_overlaps, single_key=self.single_key,
            unigrams_ignore_free_places=self.unigrams_ignore_free_places, first_stage_only=self.first_stage_only,
            defer=defer, keep=keep)

    def _aggregate_params(self):
        return dict(
            max_occurrences_1=self.max_hits, n_docs_complete_score=self.fully_score, alpha=self.score_exponent,
            beta=self.repetition_penalty, length_penalty=self.scoring_length_penalty,
            use_fm_index_frequency=self.use_fm_index_frequency,
            add_best_unigrams_to_ngrams=self.add_best_unigrams_to_ngrams, use_top_k_unigrams=self.use_top_k_ngrams,
            sort_by_length=self.sort_by_length, sort_by_freq=self.sort_by_freq, smoothing=self.smoothing,
            allow_overlaps=self.allow_overlaps, single_key=self.single_key,
            unigrams_ignore_free_places=self.unigrams_ignore_free_places, first_stage_only=self.first_stage_only)

    def batch_retrieve_from_keys(self, keys, keep=None):
        """Queries are aggregated a chunk (``batch_size``) at a time with the index work batched across
        the chunk (``aggregate_evidence_batch``).  With ``jobs >= 2`` and ``first_stage_only`` the
        host bookkeeping of each query runs in a worker process while this thread keeps the GPU busy
        with the next chunk (the producer/consumer overlap of the reference's ``Pool.imap``,
        retrieval.py:766)."""
        params = self._aggregate_params()
        on_gpu = self.gpu_aggregate and rk.gpu_aggregation_applies(self.fm_index, params)
        # worker processes only serve the host routines; on the GPU path the aggregation of a chunk is a handful of
        # launches on the index's side stream
        defer = self._host_pool() if (self.jobs >= 2 and not on_gpu) else None
        pending = []
        # with jobs >= 2 the chunk's aggregation (key scoring on the host, index kernels on their own stream; both
        # release the GIL) runs on one background thread, so that pulling the next chunk of keys -- the decode of the
        # next batch -- starts right away
        bg = self._agg_thread() if self.jobs >= 2 else None
        for chunk in _chunks(keys, self.batch_size):
            jobs = []
            for kk in chunk:
                if isinstance(kk, tuple) and len(kk) == 1:
                    kk = (kk[0], None)
                elif isinstance(kk, tuple) and len(kk) >= 2:
                    kk = (kk[0], kk[1])
                else:
                    kk = (kk, None)
                jobs.append(kk)
            if bg is not None:
                pending.append(bg.submit(rk.aggregate_evidence_batch, jobs, self.fm_index, defer=defer, keep=keep,
                                         gpu_aggregate=self.gpu_aggregate, want_ngrams=False, **params))
                continue
            yield from rk.aggregate_evidence_batch(jobs, self.fm_index, defer=defer, keep=keep, gpu_aggregate=self.gpu_aggregate,
                                                   want_ngrams=False, **params)
        import os, sys, time
        tm = rk.AGG_TIMING
        t_end = time.perf_counter()
        for fut in pending:
            t0 = time.perf_counter()
            batch_out = fut.result()
            t1 = time.perf_counter()
            for res, ngrams in batch_out:
                yield (res.result() if hasattr(res, "result") else res), ngrams
            if tm:
                print("[agg] after the last chunk was decoded: waited %.1f ms for the aggregation thread, %.1f ms for the workers of a chunk; %.1f ms since the decode ended"
                      % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3, (time.perf_counter() - t_end) * 1e3), file=sys.stderr, flush=True)

    def _agg_thread(self):
        if getattr(self, "_agg_pool", None) is None:
            import sys
            from concurrent.futures import ThreadPoolExecutor
            self._agg_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="seal-aggregate")
            # the decode loop on the main thread feeds the GPU a ~3 ms step at a time; with CPython's
            # default 5 ms switch interval every hand-over of the GIL from the aggregation thread
            # starves the GPU for up to two steps
            if sys.getswitchinterval() > 5e-4:
                sys.setswitchinterval(5e-4)
        return self._agg_pool

    def doc(self, docid: Union[str, int]) -> Optional[SEALDocument]:
        idx = self.docid2idx[docid] if isinstance(docid, str) else docid
        return SEALDocument(idx, None, self.fm_index, self.bart_tokenizer, delim1=self.title_eos_token_id,
                            delim2=self.code_eos_token_id)
