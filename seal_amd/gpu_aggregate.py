"""Evidence aggregation on the GPU: the first stage and the full-document scoring of
``aggregate_evidence`` (reference seal/keys.py:311-497) for a chunk of queries through
``fmi_agg_pack`` + ``fmi_dev_aggregate`` (seal_amd/csrc/fmi_aggregate.hip).

The host keeps what the reference does per KEY (counts -> log-odds scores in libm float64, the
rare/frequent split, the sorted dicts: keys.py:207-309, a few thousand items per query); everything it
does per located ROW and per candidate DOCUMENT (10^5-10^6 rows, 1500 documents x ~140 tokens per query)
runs on the device, and only the caller's top-k documents come back: score, accepted keys, best key and
the document tokens.  ``fmi_first_stage`` / ``fmi_full_score`` (host C++) remain as the bit-exact checkers
and as the fall-back for a query that exceeds a device limit.
"""
import ctypes
import os
from collections.abc import Mapping
from itertools import chain
from typing import List, Optional

import numpy as np

from ._lib import check, lib


def _h2d(arr, dev):
    from .keys import _h2d as f          # staged through pinned memory: does not block the host (seal_amd/keys.py)
    return f(arr, dev)

MAX_QUERIES_PER_PLAN = 256
MAX_TOP = 8192
MAX_DOC_LEN = 8192
MAX_KEY_LEN = 255


def gpu_aggregation_applies(index, params) -> bool:
    """the device path covers the searcher's configuration: ranking by score (not ``sort_by_length`` /
    ``sort_by_freq``), complete search; the index must be a resident HIP index"""
    if os.environ.get("SEAL_HOST_AGGREGATE") == "1":
        return False
    if not hasattr(index, "handle") or params.get("first_stage_only") or params.get("sort_by_length") or params.get("sort_by_freq"):
        return False
    n_top = int(params.get("n_docs_complete_score", 500))
    if not (1 <= n_top <= MAX_TOP):
        return False
    longest = index.__dict__.get("_max_doc_len")
    if longest is None:
        b = np.asarray(index.beginnings, dtype=np.int64)
        longest = index.__dict__["_max_doc_len"] = int(np.diff(b).max()) if len(b) > 1 else 0
    return 1 <= longest <= MAX_DOC_LEN


class _Buffers:
    """device workspace / output and pinned staging buffers of an index, grown on demand and reused"""

    def __init__(self):
        self.ws = self.out = self.blob = self.pin_in = self.pin_out = None

    @staticmethod
    def _fit(t, nbytes, **kw):
        import torch
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, **kw)
        return t

    def fit(self, dev, ws_bytes, out_bytes, blob_bytes, fetch_bytes):
        self.ws = self._fit(self.ws, ws_bytes, device=dev)
        self.out = self._fit(self.out, out_bytes, device=dev)
        self.blob = self._fit(self.blob, blob_bytes, device=dev)
        self.pin_in = self._fit(self.pin_in, blob_bytes, pin_memory=True)
        self.pin_out = self._fit(self.pin_out, fetch_bytes, pin_memory=True)


class _LazyPicks:
    """``[(ngram, score), ...]`` of one document, built on first use"""
    __slots__ = ("_ids", "_sc", "_keys", "_list")

    def __init__(self, ids, sc, keys):
        self._ids, self._sc, self._keys, self._list = ids, sc, keys, None

    def _get(self):
        if self._list is None:
            self._list = [((self._keys[i] if i >= 0 else (-i - 1,)), s) for i, s in zip(self._ids.tolist(), self._sc.tolist())]
        return self._list

    def __len__(self):
        return len(self._ids)

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, i):
        return self._get()[i]

    def __eq__(self, other):
        return self._get() == list(other)

    def __repr__(self):
        return repr(self._get())


class _LazyTokens:
    __slots__ = ("_arr", "_list")

    def __init__(self, arr):
        self._arr, self._list = arr, None

    def _get(self):
        if self._list is None:
            self._list = self._arr.tolist()
        return self._list

    def __len__(self):
        return len(self._arr)

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, i):
        return self._get()[i]

    def __eq__(self, other):
        return self._get() == list(other)

    def __repr__(self):
        return repr(self._get())

    def index(self, *a):
        return self._get().index(*a)

    def count(self, x):
        return self._get().count(x)


def aggregate_on_gpu(index, requests, params) -> List[Optional[dict]]:
    """``requests``: the ("locate", los, his, max_hits, ctx) requests of the queries of a chunk (seal_amd/keys.py
    ``_aggregate_steps``).  Returns per query the ``results`` dict of ``aggregate_evidence`` -- doc ->
    ``[score, [(ngram, score)...], None, doc_tokens, [best_ngram, best_score]]`` by descending score, cut to ``keep``
    -- or None where the query has to go through the host routines."""
    out: List[Optional[dict]] = []
    for a in range(0, len(requests), MAX_QUERIES_PER_PLAN):
        out += _aggregate_plan(index, requests[a:a + MAX_QUERIES_PER_PLAN], params)
    return out


def _plan_header(blob_ptr):
    """the FmiAggHeader fields of a plan blob (seal_amd/csrc/fmi_agg.h) as a dict"""
    import struct
    names = ["magic", "bytes", "nq", "n_keys", "n_rare", "total_occ", "vocab", "max_key_len", "max_u", "max_q_keys", "n_uni",
             "n_trie_slots", "n_tok", "o_q_key_off", "o_q_rare_off", "o_rare_key", "o_rare_occ_off", "o_key_lo"]
    raw = ctypes.string_at(blob_ptr, 8 * len(names))
    return dict(zip(names, struct.unpack("<%dQ" % len(names), raw)))


def _run_plan(index, plan, nq, params, two_phase=False):
    """copies the plan to the GPU, runs ``fmi_dev_aggregate`` on the index's retrieval stream and fetches the records of
    the top documents.  Returns (keep, n_out, flags, records..., pools...) as numpy arrays.  ``two_phase``: returns once the
    launches and the first copy back are ENQUEUED -- a callable that waits for them and fetches the rest (the plan and the
    index's aggregation buffers belong to this call until it has run)."""
    import torch
    from .index import SHIFT
    L = lib()
    n_top = int(params.get("n_docs_complete_score", 500))
    keep = params.get("keep")
    keep = n_top if keep is None else max(1, min(int(keep), n_top))
    blob_bytes = ctypes.c_uint64()
    blob_ptr = L.fmi_agg_plan_blob(plan, ctypes.byref(blob_bytes))
    blob_bytes = int(blob_bytes.value)
    ws_bytes = ctypes.c_uint64()
    layout = (ctypes.c_uint64 * 20)()
    allow = int(bool(params.get("allow_overlaps", False)))
    check(L.fmi_dev_aggregate_sizes(index.handle, plan, n_top, keep, allow, ctypes.byref(ws_bytes), layout))
    off = list(layout)
    fixed_bytes, out_bytes = off[17], off[18]
    dev = torch.device("cuda", L.fmi_device(index.handle))
    bufs = index.__dict__.get("_agg_buffers")
    if bufs is None:
        bufs = index.__dict__["_agg_buffers"] = _Buffers()
    R = nq * keep
    up_bytes = (blob_bytes + 15) & ~15
    bufs.fit(dev, ws_bytes.value, out_bytes, up_bytes, max(fixed_bytes, 1 << 20))
    st = index._side_stream(dev)
    with torch.cuda.stream(st):
        ctypes.memmove(bufs.pin_in.data_ptr(), blob_ptr, blob_bytes)
        # (by a kernel, not hipMemcpyAsync: the DMA queue holds the pending copy-back of a decode enqueued ahead, and a copy of this size
        # waited behind it until that decode had ended -- fmi_upload.hip)
        check(L.fmi_dev_kernel_copy(st.cuda_stream, bufs.blob.data_ptr(), bufs.pin_in.data_ptr(), up_bytes))
        mk = index.__dict__.get("_agg_mark")                  # (SEAL_OVERLAP_TIMING=2: events on this stream, retrieval.py)
        if mk is not None:
            mk("  index stream: plan of %d bytes on the GPU" % blob_bytes, st)
        check(L.fmi_dev_aggregate(
            index.handle, st.cuda_stream, plan, bufs.blob.data_ptr(), n_top, keep, allow, float(params.get("beta", 0.8)),
            float(params.get("single_key", 0.0)), int(bool(params.get("single_key_add_unigrams", False))),
            int(bool(params.get("unigrams_ignore_free_places", False))), SHIFT, bufs.ws.data_ptr(), bufs.ws.numel(),
            bufs.out.data_ptr(), bufs.out.numel()))
        check(L.fmi_dev_kernel_copy(st.cuda_stream, bufs.pin_out.data_ptr(), bufs.out.data_ptr(), (fixed_bytes + 3) & ~3))   # (likewise)

    def fetch():
        with torch.cuda.stream(st):
            return _fetch_records(index, plan, nq, params, bufs, st, off, fixed_bytes, keep, n_top, R, blob_ptr)
    return fetch if two_phase else fetch()


def _fetch_records(index, plan, nq, params, bufs, st, off, fixed_bytes, keep, n_top, R, blob_ptr):
    """the second half of ``_run_plan``: waits for the launches, reads the fixed records, then the used prefixes of the pools"""
    st.synchronize()
    fixed = bufs.pin_out[:fixed_bytes].numpy().copy()
    view = lambda slot, dt, n: fixed[off[slot]:off[slot] + n * np.dtype(dt).itemsize].view(dt)
    cursor = view(2, np.uint32, 2)
    n_picks, n_toks = int(cursor[0]), int(cursor[1])
    # the used prefixes of the pick and token pools, one more round trip
    need = 4 * n_picks + 8 * n_picks + 4 * n_toks + 64
    bufs.pin_out = bufs._fit(bufs.pin_out, need, pin_memory=True)
    a0, a1, a2 = 0, (4 * n_picks + 15) & ~15, ((4 * n_picks + 15) & ~15) + 8 * n_picks
    L = lib()
    po, do = bufs.pin_out.data_ptr(), bufs.out.data_ptr()
    if n_picks:
        check(L.fmi_dev_kernel_copy(st.cuda_stream, po + a0, do + off[12], 4 * n_picks))
        check(L.fmi_dev_kernel_copy(st.cuda_stream, po + a1, do + off[13], 8 * n_picks))
    if n_toks:
        check(L.fmi_dev_kernel_copy(st.cuda_stream, po + a2, do + off[14], 4 * n_toks))
    st.synchronize()
    host = bufs.pin_out.numpy()
    out = dict(keep=keep, n_out=view(0, np.uint32, nq), flags=view(1, np.uint32, nq),
               pick_id=host[a0:a0 + 4 * n_picks].view(np.int32).copy(), pick_score=host[a1:a1 + 8 * n_picks].view(np.float64).copy(),
               tokens=host[a2:a2 + 4 * n_toks].view(np.int32).copy(),
               doc=view(3, np.uint64, R), score=view(4, np.float64, R), best_score=view(5, np.float64, R),
               best_key=view(6, np.int32, R), T=view(7, np.uint32, R), npicks=view(8, np.uint32, R),
               pick_off=view(9, np.uint32, R), tok_off=view(10, np.uint32, R))
    tracing = getattr(index, "_trace", None) is not None
    if index.__dict__.get("_agg_debug") is not None or tracing:       # tests / bench.py parity: the first-stage ranking as well
        fs_cnt = view(11, np.uint32, nq).copy()
        fs_doc = bufs.out[off[15]:off[15] + 4 * nq * n_top].cpu().numpy().view(np.uint32).reshape(nq, n_top)
        fs_score = bufs.out[off[16]:off[16] + 8 * nq * n_top].cpu().numpy().view(np.float64).reshape(nq, n_top)
        if index.__dict__.get("_agg_debug") is not None:
            index.__dict__["_agg_debug"].append([(fs_doc[q, :fs_cnt[q]].copy(), fs_score[q, :fs_cnt[q]].copy()) for q in range(nq)])
        if tracing:
            # bench.py records every index operation of a batch with the GPU's answer to compare them with the CPU
            # oracle's: the located rows and the candidate documents never reach the host on this path, so the same
            # rows / documents are fetched once more through the host-visible calls (which record)
            H = _plan_header(blob_ptr)
            rare_key = np.frombuffer(ctypes.string_at(blob_ptr + H["o_rare_key"], 4 * H["n_rare"]), dtype=np.uint32)
            occ = np.frombuffer(ctypes.string_at(blob_ptr + H["o_rare_occ_off"], 8 * (H["n_rare"] + 1)), dtype=np.uint64).astype(np.int64)
            key_lo = np.frombuffer(ctypes.string_at(blob_ptr + H["o_key_lo"], 8 * max(H["n_keys"], 1)), dtype=np.uint64).astype(np.int64)
            lo = key_lo[rare_key.astype(np.int64)] if H["n_rare"] else np.zeros(0, np.int64)
            index.locate_ranges(lo, lo + np.diff(occ), int(params.get("max_occurrences_1", 1500)))
            index.get_docs_batch(np.concatenate([fs_doc[q, :fs_cnt[q]] for q in range(nq)]).astype(np.int64), as_arrays="flat")
    return out


class _Results(Mapping):
    """``results`` of one query from the fetched records: doc -> [score, picks, None, tokens, [best ngram, best score]] by
    descending score, as ``aggregate_evidence`` returns it -- a read-only mapping over the fetched arrays whose entries are
    built when somebody asks for them (the searcher reads ids, scores and token slices of the top k straight from the arrays:
    ``top``; 2 000 five-element entries with their lazy members per batch were 9 ms of host time nobody looked at).

    Contract (both aggregation paths): ``results`` is a ``collections.abc.Mapping`` in ranked order -- ``len``, iteration,
    ``items()`` / ``keys()`` / ``values()``, ``[doc]``, ``in``, ``itertools.islice(results.items(), k)`` work on the host path's
    ``dict`` and on this object alike; code that wants to mutate or serialise takes ``to_dict()`` (here) / the dict itself (host)."""

    def __init__(self, out, qi, k0, ngram_of):
        keep = out["keep"]
        a, n = qi * keep, int(out["n_out"][qi])
        self._out, self._a, self._n, self._k0, self._ngram_of = out, a, n, k0, ngram_of
        self._docs = out["doc"][a:a + n].tolist()
        self._pos = None
        self._entries = {}

    def __len__(self):
        return self._n

    def __iter__(self):
        return iter(self._docs)

    def _entry(self, x):
        e = self._entries.get(x)
        if e is None:
            out, a, k0 = self._out, self._a + x, self._k0
            po, npk, to, T = int(out["pick_off"][a]), int(out["npicks"][a]), int(out["tok_off"][a]), int(out["T"][a])
            ids = out["pick_id"][po:po + npk]
            ids = np.where(ids >= 0, ids - k0, ids)                 # table key id -> index into this query's table
            bk = int(out["best_key"][a])
            e = self._entries[x] = [float(out["score"][a]), _LazyPicks(ids, out["pick_score"][po:po + npk], self._ngram_of), None,
                                    _LazyTokens(out["tokens"][to:to + T]),
                                    [self._ngram_of[bk - k0] if bk >= 0 else [], float(out["best_score"][a])]]
        return e

    def __getitem__(self, doc):
        if self._pos is None:
            self._pos = {d: x for x, d in enumerate(self._docs)}
        return self._entry(self._pos[doc])

    def items(self):
        return _ItemsOf(self)

    def to_dict(self) -> dict:
        """a plain ``dict`` (doc -> entry, ranked order) with lists for the picks and the tokens: what the host path returns"""
        return {d: [e[0], list(e[1]), e[2], list(e[3]), list(e[4])] for d, e in ((d, self._entry(x)) for x, d in enumerate(self._docs))}

    def top(self, k):
        """(doc ids, scores, token arrays) of the first ``k`` documents, no entry built"""
        out, a = self._out, self._a
        n = min(self._n, k)
        to, T = out["tok_off"][a:a + n].tolist(), out["T"][a:a + n].tolist()
        toks = out["tokens"]
        return self._docs[:n], out["score"][a:a + n].tolist(), [_LazyTokens(toks[o:o + t]) for o, t in zip(to, T)]


class _ItemsOf:
    """``dict.items()`` of a ``_Results``: ordered (doc, entry) pairs, sized, iterable any number of times"""

    def __init__(self, res):
        self._res = res

    def __len__(self):
        return len(self._res)

    def __iter__(self):
        r = self._res
        return ((d, r._entry(x)) for x, d in enumerate(r._docs))


def _results_of(out, qi, k0, ngram_of):
    return _Results(out, qi, k0, ngram_of)


def _aggregate_plan(index, requests, params):
    """the python-scored keys of the queries (``_aggregate_steps`` requests) -> fmi_agg_pack -> device"""
    L = lib()
    nq = len(requests)
    max_hits = int(requests[0][3])
    # ---- table keys: the positive keys of all_ngrams, in its order; rare ones carry their row range ----
    table: List[list] = []
    q_key_off = np.zeros(nq + 1, dtype=np.int64)
    lens, scores, rare_flags, los, his, type_ptrs, keep_alive = [], [], [], [], [], [], []
    vocab = 1
    for qi, req in enumerate(requests):
        ctx = req[4]
        rare, all_ngrams, us = ctx["rare"], ctx["all_ngrams"], ctx["unigram_scores"]
        rng = dict(zip(rare.keys(), zip(np.asarray(req[1]).tolist(), np.asarray(req[2]).tolist())))
        keys = [k for k, sc in all_ngrams.items() if sc > 0.0 and len(k) >= 1]
        if any(len(k) > MAX_KEY_LEN for k in keys):
            return [None] * nq
        table.append(keys)
        q_key_off[qi + 1] = q_key_off[qi] + len(keys)
        lens += [len(k) for k in keys]
        scores += [all_ngrams[k] for k in keys]
        for k in keys:
            r = rng.get(k)
            rare_flags.append(r is not None)
            los.append(r[0] if r else 0)
            his.append(r[1] if r else 0)
        if us is not None:
            us = np.ascontiguousarray(us, dtype=np.float64)
            keep_alive.append(us)
            type_ptrs.append(us.ctypes.data)
            vocab = max(vocab, us.shape[0])
        else:
            type_ptrs.append(None)
    if len({u.shape[0] for u in keep_alive}) > 1:          # one vocabulary per plan
        return [None] * nq
    nk = int(q_key_off[-1])
    key_tok_off = np.zeros(nk + 1, dtype=np.int64)
    if nk:
        np.cumsum(lens, out=key_tok_off[1:])
    key_toks = (np.fromiter(chain.from_iterable(chain.from_iterable(table)), dtype=np.int64, count=int(key_tok_off[-1]))
                if nk else np.zeros(1, np.int64))
    key_score = np.asarray(scores if nk else [0.0], dtype=np.float64)
    key_rare = np.asarray(rare_flags if nk else [0], dtype=np.uint8)
    key_lo = np.asarray(los if nk else [0], dtype=np.uint64)
    key_hi = np.asarray(his if nk else [0], dtype=np.uint64)
    tp = (ctypes.c_void_p * nq)(*type_ptrs)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    plan = ctypes.c_void_p()
    check(L.fmi_agg_pack(nq, p(q_key_off), p(key_tok_off), p(key_toks), p(key_score), p(key_rare), p(key_lo), p(key_hi), max_hits,
                         int(index.size()), tp, vocab, ctypes.byref(plan)))
    try:
        out = _run_plan(index, plan, nq, params)
    finally:
        L.fmi_agg_plan_free(plan)
    return [None if out["flags"][qi] & 1 else _results_of(out, qi, int(q_key_off[qi]), table[qi]) for qi in range(nq)]


# ---------------------------------------------------------------------------
# the searcher's path: key scoring in C++ as well (fmi_agg_score_pack)
# ---------------------------------------------------------------------------
class _NgramTable:
    """table key index -> ngram tuple of one query (built on demand from the query's input keys)"""
    __slots__ = ("_src", "_keys")

    def __init__(self, src, keys):
        self._src, self._keys = src, keys

    def __getitem__(self, i):
        s = int(self._src[i])
        return tuple(self._keys[s]) if s >= 0 else (-s - 1,)


def score_and_aggregate_on_gpu(index, jobs, params, want_ngrams=True, two_phase=False):
    """``aggregate_evidence`` for the queries of a chunk with the key scoring (keys.py:207-309) in C++
    (``fmi_agg_score_pack``) and everything after it on the GPU.  ``jobs`` = [(ngrams_and_scores, unigram_scores)].
    Returns a list of ``(results, all_ngrams)`` -- ``all_ngrams`` None unless ``want_ngrams`` -- with ``None`` in place of
    a pair where the query must take the python route (device limit exceeded), or None for the whole chunk when the
    input is outside what the C++ scorer takes (an empty key).

    ``two_phase``: returns a callable instead, as soon as the key scoring is done and the aggregation's launches are ENQUEUED; calling it
    waits for them and returns the above.  The searcher's overlapped loop enqueues the next batch's rescoring forward between the two
    (the index kernels run beside the decode that is on the GPU at that time; the host does not sit waiting for them)."""
    import torch
    from .index import SHIFT
    from .keys import _unigram_ranges
    L = lib()
    nq = len(jobs)
    if two_phase:
        now = lambda value: (lambda: value)
    else:
        now = lambda value: value
    if nq > MAX_QUERIES_PER_PLAN:
        out = []
        for a in range(0, nq, MAX_QUERIES_PER_PLAN):
            part = score_and_aggregate_on_gpu(index, jobs[a:a + MAX_QUERIES_PER_PLAN], params, want_ngrams)
            if part is None:
                return now(None)
            out += part
        return now(out)
    key_lists, lens, lm, type_ptrs, keep_alive = [], [], [], [], []
    q_key_off = np.zeros(nq + 1, dtype=np.int64)
    vocab = 1
    for qi, (nas, us) in enumerate(jobs):
        keys = [k.tolist() if isinstance(k, torch.Tensor) else k for k, _ in nas]
        key_lists.append(keys)
        ln = [len(k) for k in keys]
        if ln and (min(ln) < 1 or max(ln) > MAX_KEY_LEN):
            return now(None)
        lens += ln
        lm += [sr for _, sr in nas]
        q_key_off[qi + 1] = q_key_off[qi] + len(keys)
        if us is not None:
            us = np.ascontiguousarray(us, dtype=np.float64)
            keep_alive.append(us)
            type_ptrs.append(us.ctypes.data)
            vocab = max(vocab, us.shape[0])
        else:
            type_ptrs.append(None)
    if len({u.shape[0] for u in keep_alive}) > 1 or (not params.get("use_fm_index_frequency", True) and any(len(k) == 0 for k in key_lists)):
        return now(None)
    nk = int(q_key_off[-1])
    key_tok_off = np.zeros(nk + 1, dtype=np.int64)
    if nk:
        np.cumsum(lens, out=key_tok_off[1:])
    ntok = int(key_tok_off[-1])
    key_toks = np.fromiter(chain.from_iterable(chain.from_iterable(key_lists)), dtype=np.int64, count=ntok) if ntok else np.zeros(1, np.int64)
    key_lm = np.asarray(lm if nk else [0.0], dtype=np.float64)
    # ---- one backward-search launch for every key of the chunk ----
    dev = torch.device("cuda", L.fmi_device(index.handle))
    if nk:
        st = torch.cuda.current_stream(dev)
        d_off = _h2d(key_tok_off, dev)
        d_tok = _h2d(key_toks, dev)
        rng = torch.empty(2, nk, dtype=torch.int64, device=dev)
        check(L.fmi_dev_get_range(index.handle, st.cuda_stream, nk, d_off.data_ptr(), d_tok.data_ptr(), SHIFT, rng[0].data_ptr(), rng[1].data_ptr()))
        rng = rng.cpu().numpy().view(np.uint64)
        key_lo, key_hi = np.ascontiguousarray(rng[0]), np.ascontiguousarray(rng[1])
        if getattr(index, "_trace", None) is not None:
            index._trace.append(("ranges", [k for keys in key_lists for k in keys], key_lo.copy(), key_hi.copy()))
    else:
        key_lo = key_hi = np.zeros(1, dtype=np.uint64)
    uni_lo, uni_hi = _unigram_ranges(index)
    tp = (ctypes.c_void_p * nq)(*type_ptrs)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    plan = ctypes.c_void_p()
    check(L.fmi_agg_score_pack(
        nq, p(q_key_off), p(key_tok_off), p(key_toks), p(key_lm), p(key_lo), p(key_hi), tp if keep_alive else None, vocab,
        p(uni_lo), p(uni_hi), len(uni_lo), float(index.beginnings[-1]), float(params.get("alpha", 2.0)),
        float(params.get("length_penalty", 0.0)), float(params.get("smoothing", 5.0)), int(bool(params.get("use_fm_index_frequency", True))),
        int(bool(params.get("add_best_unigrams_to_ngrams", False))), int(params.get("use_top_k_unigrams", 1000)),
        int(params.get("max_occurrences_1", 1500)), int(params.get("max_occurrences_2", 10_000_000)), int(index.size()), ctypes.byref(plan)))
    fetch = None
    try:
        nt = int(L.fmi_agg_plan_table_src(plan, None))
        table_src = np.zeros(max(nt, 1), dtype=np.int64)
        L.fmi_agg_plan_table_src(plan, p(table_src))
        H = _plan_header(L.fmi_agg_plan_blob(plan, None))
        q_tab = np.frombuffer(ctypes.string_at(L.fmi_agg_plan_blob(plan, None) + H["o_q_key_off"], 4 * (nq + 1)), dtype=np.uint32).astype(np.int64)
        ngrams = None
        if want_ngrams:
            ngrams = []
            for qi in range(nq):
                n = int(L.fmi_agg_plan_ngrams(plan, qi, None, None, None))
                src, sc = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.float64)
                L.fmi_agg_plan_ngrams(plan, qi, p(src), p(sc), None)
                keys = key_lists[qi]
                ngrams.append({(tuple(keys[s]) if s >= 0 else (-s - 1,)): v for s, v in zip(src[:n].tolist(), sc[:n].tolist())})
        if H["total_occ"] == 0 and not H["n_rare"]:
            # no query has a rare key: nothing to locate, every result is empty (keys.py:311 never iterates)
            return now([({}, None if ngrams is None else ngrams[qi]) for qi in range(nq)])
        fetch = _run_plan(index, plan, nq, params, two_phase=True)
    finally:
        if fetch is None:
            L.fmi_agg_plan_free(plan)

    def finish():
        try:
            out = fetch()
        finally:
            L.fmi_agg_plan_free(plan)
        res = []
        for qi in range(nq):
            if out["flags"][qi] & 1:
                res.append(None)
                continue
            k0 = int(q_tab[qi])
            table = _NgramTable(table_src[k0:int(q_tab[qi + 1])], key_lists[qi])
            res.append((_results_of(out, qi, k0, table), None if ngrams is None else ngrams[qi]))
        return res
    return finish if two_phase else finish()
