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
from itertools import chain
from typing import List, Optional

import numpy as np

from ._lib import check, lib

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


def _aggregate_plan(index, requests, params):
    import torch
    from .index import SHIFT
    L = lib()
    nq = len(requests)
    n_top = int(params.get("n_docs_complete_score", 500))
    keep = params.get("keep")
    keep = n_top if keep is None else max(1, min(int(keep), n_top))
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
        bufs.fit(dev, ws_bytes.value, out_bytes, blob_bytes, max(fixed_bytes, 1 << 20))
        st = index._side_stream(dev)
        with torch.cuda.stream(st):
            ctypes.memmove(bufs.pin_in.data_ptr(), blob_ptr, blob_bytes)
            bufs.blob[:blob_bytes].copy_(bufs.pin_in[:blob_bytes], non_blocking=True)
            check(L.fmi_dev_aggregate(
                index.handle, st.cuda_stream, plan, bufs.blob.data_ptr(), n_top, keep, allow, float(params.get("beta", 0.8)),
                float(params.get("single_key", 0.0)), int(bool(params.get("single_key_add_unigrams", False))),
                int(bool(params.get("unigrams_ignore_free_places", False))), SHIFT, bufs.ws.data_ptr(), bufs.ws.numel(),
                bufs.out.data_ptr(), bufs.out.numel()))
            bufs.pin_out[:fixed_bytes].copy_(bufs.out[:fixed_bytes], non_blocking=True)
            st.synchronize()
            fixed = bufs.pin_out[:fixed_bytes].numpy().copy()
            view = lambda slot, dt, n: fixed[off[slot]:off[slot] + n * np.dtype(dt).itemsize].view(dt)
            n_out, flags, cursor = view(0, np.uint32, nq), view(1, np.uint32, nq), view(2, np.uint32, 2)
            n_picks, n_toks = int(cursor[0]), int(cursor[1])
            # the used prefixes of the pick and token pools, one more round trip
            need = 4 * n_picks + 8 * n_picks + 4 * n_toks + 64
            bufs.pin_out = bufs._fit(bufs.pin_out, need, pin_memory=True)
            a0, a1, a2 = 0, (4 * n_picks + 15) & ~15, ((4 * n_picks + 15) & ~15) + 8 * n_picks
            if n_picks:
                bufs.pin_out[a0:a0 + 4 * n_picks].copy_(bufs.out[off[12]:off[12] + 4 * n_picks], non_blocking=True)
                bufs.pin_out[a1:a1 + 8 * n_picks].copy_(bufs.out[off[13]:off[13] + 8 * n_picks], non_blocking=True)
            if n_toks:
                bufs.pin_out[a2:a2 + 4 * n_toks].copy_(bufs.out[off[14]:off[14] + 4 * n_toks], non_blocking=True)
            st.synchronize()
            host = bufs.pin_out.numpy()
            pick_id = host[a0:a0 + 4 * n_picks].view(np.int32).copy()
            pick_score = host[a1:a1 + 8 * n_picks].view(np.float64).copy()
            tokens = host[a2:a2 + 4 * n_toks].view(np.int32).copy()
            if index.__dict__.get("_agg_debug") is not None:       # tests: the first-stage ranking as well
                fs_cnt = view(11, np.uint32, nq).copy()
                fs_doc = bufs.out[off[15]:off[15] + 4 * nq * n_top].cpu().numpy().view(np.uint32).reshape(nq, n_top)
                fs_score = bufs.out[off[16]:off[16] + 8 * nq * n_top].cpu().numpy().view(np.float64).reshape(nq, n_top)
                index.__dict__["_agg_debug"].append([(fs_doc[q, :fs_cnt[q]].copy(), fs_score[q, :fs_cnt[q]].copy()) for q in range(nq)])
    finally:
        L.fmi_agg_plan_free(plan)
    rec_doc, rec_score, rec_best_score = view(3, np.uint64, R), view(4, np.float64, R), view(5, np.float64, R)
    rec_best_key, rec_T, rec_np = view(6, np.int32, R), view(7, np.uint32, R), view(8, np.uint32, R)
    rec_po, rec_to = view(9, np.uint32, R), view(10, np.uint32, R)
    results: List[Optional[dict]] = []
    for qi in range(nq):
        if flags[qi] & 1:
            results.append(None)
            continue
        keys = table[qi]
        k0 = int(q_key_off[qi])
        res = {}
        a = qi * keep
        docs = rec_doc[a:a + n_out[qi]].tolist()
        sc = rec_score[a:a + n_out[qi]].tolist()
        bk = rec_best_key[a:a + n_out[qi]].tolist()
        bs = rec_best_score[a:a + n_out[qi]].tolist()
        for x, d in enumerate(docs):
            r = a + x
            po, npk, to, T = int(rec_po[r]), int(rec_np[r]), int(rec_to[r]), int(rec_T[r])
            ids = pick_id[po:po + npk]
            ids = np.where(ids >= 0, ids - k0, ids)             # table key id -> index into this query's key list
            res[d] = [sc[x], _LazyPicks(ids, pick_score[po:po + npk], keys), None, _LazyTokens(tokens[to:to + T]),
                      [keys[bk[x] - k0] if bk[x] >= 0 else [], bs[x]]]
        results.append(res)
    return results
