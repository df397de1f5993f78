"""Key scoring and evidence aggregation (reference seal/keys.py) on the MI355X engine.

What changes with respect to the reference is *where the index work happens*,
not what is computed:

* every ``index.get_count(ngram)`` of a query (keys.py:212,252,277,287) is served
  from ONE batched backward-search launch per call of ``aggregate_evidence``
  (plus a per-index cached table for single-token counts);
* the ``locate`` + ``get_doc_index`` pair issued per matching row
  (keys.py:320-324, up to ``max_hits`` rows per rare key) becomes one launch over
  all rows of all rare keys (suffix-array gather + doc binning on the GPU);
* the documents that are fully scored are fetched in one launch
  (``get_doc``, keys.py:388).

The order-sensitive bookkeeping of the reference (dict insertion orders, stable
sorts, first-come coverage, the ``[tok_end - len, tok_end)`` window, heap order
of the greedy matcher) is reproduced on the host from those arrays; floating
point is float64 ``math`` exactly where the reference uses it.
"""
import itertools
import math
import os
from collections import Counter
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


# Rescoring forms.  Module attributes, not environment switches (round 6): the tests flip them to run the other path.
RESCORE_TREE = True        # teacher forcing over the prefix TREE of a query's keys (every distinct prefix one decoder position); False: one row per maximal parent
RESCORE_GRAPH = True       # the tree forward as one hipGraph replay (BartStepDecoder.tree_hidden_graph)
RESCORE_CHUNK_ROWS = 256   # rows per forward of the maximal-parent form
RESCORE_MAX_NODES = 48000  # nodes of one forest (a searcher batch's rescorings are ONE forward up to here)
RESCORE_LM_HEAD_SLICE = 4096   # nodes per slice of the output projection (the logits of all nodes never exist at once)
AGG_TIMING = False         # tools: aggregate_evidence_batch prints where its host time goes

def deduplicate(list_of_lists):
    """First occurrence wins; elements are token lists or (score, tokens) pairs
    (reference keys.py:19-35)."""
    seen = set()
    kept = []
    for item in list_of_lists:
        toks = item[1] if isinstance(item[0], float) else item
        key = tuple(toks.tolist()) if isinstance(toks, torch.Tensor) else tuple(toks)
        if key not in seen:
            seen.add(key)
            kept.append(item)
    return kept


def strip(seq, symbols_start, symbols_end):
    """Drop leading symbols in ``symbols_start`` and trailing ones in
    ``symbols_end`` (reference keys.py:54-61)."""
    a, b = 0, len(seq)
    while a < b and seq[a] in symbols_start:
        a += 1
    while b > a and seq[b - 1] in symbols_end:
        b -= 1
    return seq[a:b]


def _h2d(arr: np.ndarray, device) -> torch.Tensor:
    """host array -> device WITHOUT waiting for the stream: a copy from pageable memory blocks the host until everything
    queued before it has run (measured: ~1.2 ms per small index tensor in the middle of a search step, i.e. a pipeline
    bubble each time), a copy from a pinned staging buffer is just another enqueued command.  The caching host allocator
    keeps the staging buffer alive until the copy has executed."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def _pad_batch(seqs: Sequence[Sequence[int]], pad: int, device) -> torch.Tensor:
    width = max(len(s) for s in seqs)
    out = np.full((len(seqs), width), pad, dtype=np.int64)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = s
    return _h2d(out, device)


@torch.inference_mode()
def rescore_keys(model, inputs, list_of_decoded, batch_size=100, length_penalty=0.0, progress_bar=False, prefix=[],
                 strip_from_bos=[], strip_from_eos=[], logit_bias=None, share_prefixes=True, encoded=None, pending=False):
    """Teacher-forced log-probability of every key given its query
    (reference keys.py:64-141): targets with id < 2 contribute 0 (keys.py:132),
    score divided by ``len(key) ** length_penalty``.

    ``share_prefixes`` (default): the keys of a query are the nodes of the beam
    search tree -- most are prefixes or siblings of one another -- and the decoder is
    causal, so the distribution after ``key[:j]`` is the same in every key that starts
    with it.  Every DISTINCT prefix is run through the model once, as one decoder
    position that attends its ancestors (``_rescore_keys_tree``; ``keys.RESCORE_TREE = False``:
    one row per maximal key, every other key reading its score off the row's cumulative
    sums, ``_rescore_keys_shared``).  Same numbers as scoring each key separately (up to
    fp32 summation order), ~20x / ~6x fewer decoder positions.  ``share_prefixes=False``
    is the reference's one-row-per-key batching."""
    if share_prefixes:
        # the prefix tree (every distinct prefix ONE decoder position); RESCORE_TREE = False: maximal parents as rows
        fn = _rescore_keys_tree if RESCORE_TREE else _rescore_keys_shared
        job = fn(model, inputs, list_of_decoded, batch_size, length_penalty, prefix, strip_from_bos, strip_from_eos, logit_bias, encoded)
        # ``pending``: everything is enqueued and nothing has waited for the GPU; ``job.result()`` reads the scores back
        # (the searcher enqueues the three rescorings of a batch back to back and reads them back afterwards)
        return job if pending else job.result()
    if pending:
        raise NotImplementedError("pending=True needs share_prefixes=True")
    cfg = model.config
    device = next(model.parameters()).device
    if inputs is None:
        batch_in = [[cfg.bos_token_id, cfg.eos_token_id]] * len(list_of_decoded)
    else:
        batch_in = [list(i) for i in inputs]
    decoded = [[x[1] if isinstance(x[0], float) else x for x in xx] for xx in list_of_decoded]
    input_ids = _pad_batch(batch_in, cfg.pad_token_id, device)
    attention_mask = (input_ids != cfg.pad_token_id).to(torch.uint8)
    enc = model.model.encoder(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
    flat = [(qi, key) for qi, keys in enumerate(decoded) for key in keys]
    out = {qi: [] for qi in range(len(decoded))}
    start = cfg.decoder_start_token_id
    for c0 in range(0, len(flat), batch_size):
        chunk = flat[c0:c0 + batch_size]
        qidx = _h2d(np.asarray([qi for qi, _ in chunk], dtype=np.int64), device)
        dec_in = []
        for _, key in chunk:
            d = [start] + list(prefix) + list(strip(list(key), strip_from_bos, strip_from_eos))
            dec_in.append(d)
        dec_ids = _pad_batch(dec_in, cfg.pad_token_id, device)
        logits = model(attention_mask=attention_mask[qidx], encoder_outputs=(enc[qidx],),
                       decoder_input_ids=dec_ids[:, :-1]).logits
        if logit_bias is not None:      # extension: per-query additive logit bias [n_queries, vocab]
            logits = logits + logit_bias[qidx][:, None, :]
        logprobs = logits.log_softmax(-1)
        tgt = dec_ids[:, 1:]
        lp = torch.gather(logprobs, -1, tgt.unsqueeze(-1)).squeeze(-1)
        lp = torch.where(tgt < 2, torch.zeros_like(lp), lp)
        lp = lp[:, len(prefix):].sum(-1).tolist()
        for (qi, key), ll in zip(chunk, lp):
            out[qi].append((ll / (len(key) ** length_penalty), list(key)))
    return [out[qi] for qi in range(len(decoded))]


@torch.inference_mode()
def _rescore_keys_shared(model, inputs, list_of_decoded, batch_size, length_penalty, prefix, strip_from_bos, strip_from_eos,
                         logit_bias, encoded=None):
    """Tree-shared teacher forcing.  score(key) = sum_j log p(key[j] | key[:j]).  The keys of a
    query form the beam-search tree; a decoder row fed with ``[start] + q`` yields, at position j,
    the distribution after ``q[:j]`` -- i.e. every term of every key whose parent ``key[:-1]`` is a
    prefix of ``q``.  Rows are therefore only the MAXIMAL PARENTS (about one per beam), and a key
    reads ``cum_q[len-1] + logp[row, len-1, key[-1]]``."""
    cfg = model.config
    device = next(model.parameters()).device
    if inputs is None:
        batch_in = [[cfg.bos_token_id, cfg.eos_token_id]] * len(list_of_decoded)
    else:
        batch_in = [list(i) for i in inputs]
    decoded = [[x[1] if isinstance(x[0], float) else x for x in xx] for xx in list_of_decoded]
    if encoded is not None:
        # (encoder states, attention mask) of exactly these inputs from an earlier pass of the same model (the searcher's
        # title decode encodes what the title rescoring would encode again, retrieval.py:157-160 vs 195)
        enc, attention_mask = encoded
    else:
        input_ids = _pad_batch(batch_in, cfg.pad_token_id, device)
        attention_mask = (input_ids != cfg.pad_token_id).to(torch.uint8)
        sd = getattr(model, "_seal_step_decoder", None)
        if sd is None:
            from .bart_decoder import BartStepDecoder
            sd = model._seal_step_decoder = BartStepDecoder(model)
        enc = sd.encode(input_ids, attention_mask)          # the encoder without HF's blocking mask check
    start, npre = cfg.decoder_start_token_id, len(prefix)
    seqs = [[tuple(list(prefix) + list(strip(list(key), strip_from_bos, strip_from_eos))) for key in keys] for keys in decoded]
    work = []           # (query, maximal parent)
    owner = []          # per query: {parent -> index into work}
    for qi, ss in enumerate(seqs):
        parents = sorted({sq[:-1] for sq in ss if len(sq) > 0})
        own, nxt = {}, None
        for p in reversed(parents):      # lexicographic order: a sequence is followed by its extensions
            if nxt is not None and len(nxt) > len(p) and nxt[:len(p)] == p:
                own[p] = own[nxt]
            else:
                own[p] = len(work)
                work.append((qi, p))
            nxt = p
        owner.append(own)
    order = sorted(range(len(work)), key=lambda i: len(work[i][1]))     # similar lengths together: less padding
    slot = {w: j for j, w in enumerate(order)}
    # keys grouped by the chunk of their owner row
    # rows per forward: few, large launches (an eager F.linear costs the host ~19 us whatever its size, and a batch of
    # queries has a few hundred maximal parents); the logits of a chunk are rows x T x vocab floats -- 0.57 GB at 256 rows x 11 positions; 1024 rows measured no faster: more padding
    chunk_rows = max(batch_size, RESCORE_CHUNK_ROWS)
    per_chunk = {}
    for qi, ss in enumerate(seqs):
        for ki, sq in enumerate(ss):
            if len(sq) > 0:
                j = slot[owner[qi][sq[:-1]]]
                per_chunk.setdefault(j // chunk_rows, []).append((qi, ki, j % chunk_rows, sq))
    scores = [[0.0] * len(ss) for ss in seqs]
    totals = []         # per chunk: (its keys, their scores on the device)
    # decoder forward: the fused step-decoder kernels when they apply (GPU, fp32), HF's module otherwise
    stepdec = getattr(model, "_seal_step_decoder", None)
    max_T = 1 + max((len(p) for _, p in work), default=0)
    prepared = None
    if enc.is_cuda:
        if stepdec is None:
            from .bart_decoder import BartStepDecoder
            stepdec = model._seal_step_decoder = BartStepDecoder(model)
        if stepdec.can_teacher_force(enc, max_T):
            prepared = stepdec.teacher_prepare(enc, attention_mask)
    for c, items in per_chunk.items():
        rows = [work[w] for w in order[c * chunk_rows:(c + 1) * chunk_rows]]
        qidx = _h2d(np.asarray([qi for qi, _ in rows], dtype=np.int64), device)
        dec_ids = _pad_batch([[start] + list(p) for _, p in rows], cfg.pad_token_id, device)
        # (row, length, last token) of every key of the chunk: one staged copy
        key_idx = _h2d(np.asarray([[r for _, _, r, _ in items], [len(sq) for _, _, _, sq in items], [sq[-1] for _, _, _, sq in items]],
                                  dtype=np.int64), device)
        if prepared is not None:
            logits = stepdec.teacher_logits(dec_ids, qidx, prepared)
        else:
            logits = model(attention_mask=attention_mask[qidx], encoder_outputs=(enc[qidx],), decoder_input_ids=dec_ids).logits
        if logit_bias is not None:
            logits = logits + logit_bias[qidx][:, None, :]
        logp = logits.log_softmax(-1)                                   # [rows, T, V]; position j: after p[:j]
        T = dec_ids.shape[1]
        if T > 1:
            tgt = dec_ids[:, 1:]
            own_lp = torch.gather(logp[:, :-1], -1, tgt.unsqueeze(-1)).squeeze(-1)
            own_lp = torch.where(tgt < 2, torch.zeros_like(own_lp), own_lp)
            cum = torch.cat([torch.zeros(len(rows), 1, dtype=torch.float64, device=device),
                             torch.cumsum(own_lp.double(), dim=-1)], dim=1)          # cum[r, m] = first m terms
        else:
            cum = torch.zeros(len(rows), 1, dtype=torch.float64, device=device)
        r_idx, n_idx, last = key_idx[0], key_idx[1], key_idx[2]
        last_lp = logp[r_idx, n_idx - 1, last].double()
        last_lp = torch.where(last < 2, torch.zeros_like(last_lp), last_lp)
        lo = torch.clamp(torch.full_like(n_idx, npre), max=cum.shape[1] - 1)
        body = cum[r_idx, n_idx - 1] - cum[r_idx, torch.minimum(lo, n_idx - 1)]
        totals.append((items, torch.where(n_idx > npre, body + last_lp, torch.zeros_like(body)).float()))
    return _PendingRescore(totals, scores, decoded, length_penalty)


def _prefix_tree(key_query, key_seqs, npre, start):
    """The distinct prefixes of the given keys (``key_seqs[k]``: a token tuple of query ``key_query[k]``, every one longer
    than ``npre``) as a forest, built level by level with numpy (a node of depth j = a distinct (parent node, token j-1) pair;
    nodes are numbered depth-major): per node its input token (``start`` for the roots), depth, query and ancestor row
    (root first, itself last, -1 beyond); per term (key k, position j >= npre) the node of ``key[:j]``, the target
    ``key[j]``, k and the column j - npre."""
    nk = len(key_seqs)
    lens = np.fromiter(map(len, key_seqs), dtype=np.int64, count=nk)
    L = int(lens.max())
    K = np.zeros((nk, L), dtype=np.int64)
    flat = np.fromiter(itertools.chain.from_iterable(key_seqs), dtype=np.int64, count=int(lens.sum()))
    K[np.repeat(np.arange(nk), lens), np.arange(len(flat)) - np.repeat(np.cumsum(lens) - lens, lens)] = flat
    base = int(flat.max()) + 1
    ids = np.full((nk, L), -1, dtype=np.int64)              # ids[k, j] = node of key[:j]
    # depth 0: one root per query
    lvl_query, ids[:, 0] = np.unique(np.asarray(key_query, dtype=np.int64), return_inverse=True)
    lvl_anc = np.full((len(lvl_query), L), -1, dtype=np.int64)
    lvl_anc[:, 0] = np.arange(len(lvl_query))
    toks, depths, queries, ancs = [np.full(len(lvl_query), start, dtype=np.int64)], [np.zeros(len(lvl_query), dtype=np.int64)], [lvl_query], [lvl_anc]
    first = 0                                               # id of the first node of the previous level
    n_nodes = len(lvl_query)
    for j in range(1, L):
        live = np.nonzero(lens > j)[0]
        if len(live) == 0:
            break
        up, inv = np.unique(ids[live, j - 1] * base + K[live, j - 1], return_inverse=True)
        parent = up // base - first                         # index into the previous level
        ids[live, j] = n_nodes + inv
        a = lvl_anc[parent]
        a[:, j] = n_nodes + np.arange(len(up))
        lvl_query, lvl_anc = lvl_query[parent], a
        toks.append(up % base); depths.append(np.full(len(up), j, dtype=np.int64)); queries.append(lvl_query); ancs.append(a)
        first, n_nodes = n_nodes, n_nodes + len(up)
    A = len(toks)                                           # deepest node + 1
    jj = np.arange(L)[None, :]
    npre = np.broadcast_to(np.asarray(npre, dtype=np.int64), (nk,))            # one scored-from position for all keys, or one per key
    tk, tj = np.nonzero((jj >= npre[:, None]) & (jj < lens[:, None]))
    return dict(tok=np.concatenate(toks), depth=np.concatenate(depths), query=np.concatenate(queries),
                anc=np.ascontiguousarray(np.concatenate(ancs)[:, :A]),
                term_node=ids[tk, tj], term_tok=K[tk, tj], term_key=tk, term_col=tj - npre[tk], width=int(max(1, (lens - npre).max())))


def _rescore_keys_tree(model, inputs, list_of_decoded, batch_size, length_penalty, prefix, strip_from_bos, strip_from_eos,
                       logit_bias, encoded=None):
    return _rescore_tree_jobs(model, [dict(inputs=inputs, keys=list_of_decoded, length_penalty=length_penalty, prefix=prefix,
                                           strip_from_bos=strip_from_bos, strip_from_eos=strip_from_eos, logit_bias=logit_bias,
                                           encoded=encoded)])[0]


@torch.inference_mode()
def rescore_keys_multi(jobs, pending=False):
    """Several ``rescore_keys(share_prefixes=True)`` calls at once: ``jobs`` = [(model, inputs, list_of_decoded, kwargs)] with
    the keyword arguments of ``rescore_keys`` (``prefix, strip_from_bos, strip_from_eos, logit_bias, encoded, length_penalty``).
    The jobs of one model share ONE forward over the prefix forest of all their queries (the searcher's body keys / query
    n-grams / titles of a batch: 3 x ~500 launches and three under-filled sets of GEMMs become one); each job's scores are
    what its own ``rescore_keys`` call returns."""
    out = [None] * len(jobs)
    if not RESCORE_TREE:                  # the maximal-parent rows, job by job (A/B, tests)
        out = [rescore_keys(job[0], job[1], job[2], pending=True, **job[3]) for job in jobs]
        return out if pending else [r.result() for r in out]
    by_model = {}
    for j, job in enumerate(jobs):
        by_model.setdefault(id(job[0]), []).append(j)
    for ids in by_model.values():
        res = _rescore_tree_jobs(jobs[ids[0]][0], [dict(inputs=jobs[j][1], keys=jobs[j][2], **jobs[j][3]) for j in ids])
        for j, r in zip(ids, res):
            out[j] = r
    return out if pending else [r.result() for r in out]


@torch.inference_mode()
def _rescore_tree_jobs(model, jobs):
    """Teacher forcing over the prefix tree.  score(key) = sum_j log p(key[j] | key[:j]) and the decoder is causal, so the
    distribution after ``key[:j]`` is the same in every key that starts with it: every DISTINCT prefix of a query's keys is
    one node = one decoder position (input token ``key[j-1]`` at position ``j``, attending its ancestors), run through the
    model once (``BartStepDecoder.tree_logits``).  The keys of a batch of 20 searcher queries -- the recorded hypotheses of
    a beam search: a few thousand keys -- have ~3 000 distinct prefixes where one row per maximal parent
    (``_rescore_keys_shared``) runs ~10 000 positions and one row per key (reference keys.py:64-141) ~60 000.

    ``jobs``: dicts with inputs / keys / prefix / strip_from_bos / strip_from_eos / logit_bias / encoded / length_penalty.
    The queries of all jobs form one forest (a job's queries keep their own encoder states: the states of the jobs are
    padded to one length and stacked); one ``_PendingRescore`` per job."""
    cfg = model.config
    device = next(model.parameters()).device
    sd = getattr(model, "_seal_step_decoder", None)
    if sd is None or sd.layers[0]["qkv_w"].device != device:        # (the fused weights follow the model to its device)
        from .bart_decoder import BartStepDecoder
        sd = model._seal_step_decoder = BartStepDecoder(model)
    start = cfg.decoder_start_token_id
    encs, masks, J = [], [], []
    for job in jobs:
        list_of_decoded = job["keys"]
        if job.get("inputs") is None:
            batch_in = [[cfg.bos_token_id, cfg.eos_token_id]] * len(list_of_decoded)
        else:
            batch_in = [list(i) for i in job["inputs"]]
        decoded = [[x[1] if isinstance(x[0], float) else x for x in xx] for xx in list_of_decoded]
        if job.get("encoded") is not None:
            # (encoder states, attention mask) of exactly these inputs from an earlier pass of the same model (the searcher's
            # decodes encode what the rescorings would encode again, retrieval.py:157-160 vs 195)
            enc, attention_mask = job["encoded"]
        else:
            input_ids = _pad_batch(batch_in, cfg.pad_token_id, device)
            attention_mask = (input_ids != cfg.pad_token_id).to(torch.uint8)
            if device.type == "cuda":
                enc = sd.encode(input_ids, attention_mask)          # the encoder without HF's blocking mask check
            else:
                enc = model.model.encoder(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
        prefix = list(job.get("prefix") or [])
        sfb, sfe = job.get("strip_from_bos") or [], job.get("strip_from_eos") or []
        seqs = [[tuple(prefix + list(strip(list(key), sfb, sfe))) for key in keys] for keys in decoded]
        encs.append(enc); masks.append(attention_mask.to(torch.uint8))
        J.append(dict(decoded=decoded, seqs=seqs, npre=len(prefix), scores=[[0.0] * len(ss) for ss in seqs], totals=[],
                      lp=job.get("length_penalty", 0.0), bias=job.get("logit_bias"), first=sum(len(x["seqs"]) for x in J)))
    if len(jobs) == 1:
        enc, attention_mask = encs[0], masks[0]
    else:
        S = max(e.shape[1] for e in encs)
        enc = torch.cat([torch.nn.functional.pad(e, (0, 0, 0, S - e.shape[1])) for e in encs])
        attention_mask = torch.cat([torch.nn.functional.pad(m, (0, S - m.shape[1])) for m in masks])
    logit_bias = None
    if any(j["bias"] is not None for j in J):
        V = next(j["bias"] for j in J if j["bias"] is not None).shape[-1]
        logit_bias = torch.cat([j["bias"] if j["bias"] is not None else torch.zeros(len(j["seqs"]), V, dtype=enc.dtype, device=device)
                                for j in J])
    # whole queries per forward; the sum of their key lengths, an upper bound of the nodes (3x to 10x: title keys share long prefixes),
    # stays below `cap`.  The cap dates from when nodes x vocab logits existed at once; they are projected a slice at a time now, and a
    # cap of 12 000 cut a searcher batch (20 queries x 3 jobs, ~3 300 nodes) into forests of ~2 000 + 700 + 650 nodes = graphs of 2 048 +
    # 1 024 + 1 024 rows, three forwards of launch-bound height.  One forest of 4 096 rows is the same padded rows in a third of the
    # launches at twice the GEMM efficiency: 332 -> 367 queries/s on one box (profiles/r5_rescore_one_forest_ab.txt).
    cap = RESCORE_MAX_NODES
    max_len = max((len(sq) for j in J for ss in j["seqs"] for sq in ss), default=0)
    prepared = None
    fused_ok = enc.is_cuda and max_len > 0 and sd.can_teacher_force(enc, max_len)
    use_graph = fused_ok and RESCORE_GRAPH      # (on by default: see BartStepDecoder.tree_hidden_graph)
    units = [(ji, qi) for ji, j in enumerate(J) for qi in range(len(j["seqs"]))]          # job-major: a job's keys stay contiguous
    groups, cur, cur_n = [], [], 0
    for ji, qi in units:
        n = sum(len(sq) for sq in J[ji]["seqs"][qi] if len(sq) > J[ji]["npre"])
        if cur and cur_n + n > cap:
            groups.append(cur); cur, cur_n = [], 0
        cur.append((ji, qi)); cur_n += n
    if cur:
        groups.append(cur)
    for group in groups:
        items = [(ji, qi, ki, sq) for ji, qi in group for ki, sq in enumerate(J[ji]["seqs"][qi]) if len(sq) > J[ji]["npre"]]
        if not items:
            continue
        tree = _prefix_tree([J[ji]["first"] + qi for ji, qi, _, _ in items], [sq for _, _, _, sq in items],
                            np.fromiter((J[ji]["npre"] for ji, _, _, _ in items), dtype=np.int64, count=len(items)), start)
        packed = _h2d(np.stack([tree["tok"], tree["depth"], tree["query"]]), device)
        anc_d = _h2d(tree["anc"], device)
        hidden = None
        if use_graph:
            hidden = sd.tree_hidden_graph(packed[0], packed[1], anc_d, packed[2], enc, attention_mask)              # one graph replay
        if hidden is None:
            if fused_ok and prepared is None:
                prepared = sd.teacher_prepare(enc, attention_mask)
            hidden = sd.tree_logits(packed[0], packed[1], anc_d, packed[2], enc, attention_mask, prepared, True)    # [nodes, d]
        # term j of key k = logp[node(key[:j]), key[j]] (0 for targets < 2, keys.py:132), summed in position order in float64.
        # The output projection + log-softmax run over a slice of the nodes at a time (the terms sorted by node, so that a
        # slice's terms are one run): nodes x vocab floats never exist at once -- 12 000 nodes x 50 265 x 4 B = 2.4 GB, three
        # times with the bias add and the log-softmax -- and every row's arithmetic is what the whole-matrix form computes.
        order = np.argsort(tree["term_node"], kind="stable")
        t_node = tree["term_node"][order]
        t = _h2d(np.stack([t_node, tree["term_tok"][order], tree["term_key"][order], tree["term_col"][order]]), device)
        table = torch.zeros(len(items), tree["width"], dtype=torch.float64, device=device)
        n_nodes = hidden.shape[0]
        step = max(256, RESCORE_LM_HEAD_SLICE)
        for a in range(0, n_nodes, step):
            b = min(n_nodes, a + step)
            ta, tb = int(np.searchsorted(t_node, a, side="left")), int(np.searchsorted(t_node, b, side="left"))
            if ta == tb:
                continue
            logits = sd.lm_head(hidden[a:b])
            if logit_bias is not None:
                logits = logits + logit_bias[packed[2][a:b]]
            logp = logits.log_softmax(-1)                               # [slice, V]: the distribution after the node's prefix
            lp = logp[t[0, ta:tb] - a, t[1, ta:tb]].double()
            lp = torch.where(t[1, ta:tb] < 2, torch.zeros_like(lp), lp)
            table[t[2, ta:tb], t[3, ta:tb]] = lp
        total = table.sum(-1).float()
        a = 0
        while a < len(items):                                           # the group's keys, job by job
            b = a
            while b < len(items) and items[b][0] == items[a][0]:
                b += 1
            J[items[a][0]]["totals"].append(([(qi, ki, 0, sq) for _, qi, ki, sq in items[a:b]], total[a:b]))
            a = b
    from . import split_gemm
    flag = split_gemm.flag_snapshot(device) if device.type == "cuda" else None     # read with the scores: activations beyond fp16's range?
    # the snapshot rides with the FIRST job that has scores to read back (a job without totals never looks at it)
    first = next((ji for ji, j in enumerate(J) if j["totals"]), None)
    return [_PendingRescore(j["totals"], j["scores"], j["decoded"], j["lp"], flag if ji == first else None, device) for ji, j in enumerate(J)]


class _PendingRescore:
    """the enqueued chunks of one ``rescore_keys`` call: one read-back for all of them"""

    def __init__(self, totals, scores, decoded, length_penalty, split_flag=None, device=None):
        self._totals, self._scores, self._decoded, self._lp = totals, scores, decoded, length_penalty
        self._split_flag, self._device = split_flag, device
        # The scores start for pinned host memory NOW, behind the forward that makes them, and an event marks their arrival: a ``.tolist()`` in
        # ``result`` would be a copy on whatever the stream holds BY THEN -- the next batch's rescoring forward, if the searcher has enqueued it
        self._host, self._ready = None, None
        if totals and totals[0][1].is_cuda:
            t = torch.cat([t for _, t in totals]) if len(totals) > 1 else totals[0][1]
            self._host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._host.copy_(t, non_blocking=True)
            self._ready = torch.cuda.Event()
            self._ready.record(torch.cuda.current_stream(t.device))

    def result(self):
        scores = self._scores
        if self._totals:
            if self._host is not None:
                self._ready.synchronize()
                flat = self._host.tolist()
            else:
                flat = torch.cat([t for _, t in self._totals]).tolist() if len(self._totals) > 1 else self._totals[0][1].tolist()
            if self._split_flag is not None:          # (the read-back above waited for the stream the snapshot was enqueued on)
                from . import split_gemm
                flag, self._split_flag = self._split_flag, None
                split_gemm.check_snapshot(flag, self._device)
            pos = 0
            for items, _ in self._totals:
                for (qi, ki, _, _), ll in zip(items, flat[pos:pos + len(items)]):
                    scores[qi][ki] = ll
                pos += len(items)
        lp = self._lp
        return [[(scores[qi][ki] / (len(key) ** lp), list(key)) for ki, key in enumerate(keys)]
                for qi, keys in enumerate(self._decoded)]


@torch.no_grad()
def compute_unigram_scores(model, inputs, index=None, tokenizer=None, tolist=True, temperature=1.0, prefix=[],
                           logit_bias=None):
    """One decoder step -> log-softmax over the vocabulary per query
    (reference keys.py:145-176)."""
    cfg = model.config
    device = next(model.parameters()).device
    if isinstance(inputs[0], str):
        batch = tokenizer(list(inputs), padding=True, return_tensors="pt")
        input_ids, attention_mask = batch["input_ids"].to(device), batch["attention_mask"].to(device)
    else:
        input_ids = _pad_batch([list(i) for i in inputs], cfg.pad_token_id, device)
        attention_mask = (input_ids != cfg.pad_token_id).to(torch.uint8)
    dec = torch.full((input_ids.shape[0], 1 + len(prefix)), cfg.decoder_start_token_id, dtype=torch.long, device=device)
    for j, tok in enumerate(prefix, start=1):
        dec[:, j] = tok
    logits = model(input_ids=input_ids, attention_mask=attention_mask, decoder_input_ids=dec).logits[:, len(prefix)]
    if logit_bias is not None:
        logits = logits + logit_bias
    if temperature != 1.0:
        logits = logits / temperature
    logprobs = logits.log_softmax(-1)
    return logprobs.tolist() if tolist else logprobs


# ---------------------------------------------------------------------------
# aggregate_evidence
# ---------------------------------------------------------------------------
def _log_odds(sr: float, count: int, ntokens: float, smoothing: float) -> float:
    """LM-vs-corpus log-odds of keys.py:221-224 in float64 ``math``."""
    snr = math.log((count + smoothing) / (ntokens + smoothing))
    return (sr + math.log(1 - math.exp(snr))) - (snr + math.log(1 - math.exp(sr)))


def _log_odds_many(sr: np.ndarray, counts: np.ndarray, ntokens: float, smoothing: float) -> np.ndarray:
    """``_log_odds`` for arrays through libsealfm (libm log/exp on doubles == python's math module)."""
    import ctypes
    from ._lib import check, lib
    sr = np.ascontiguousarray(sr, dtype=np.float64)
    cnt = np.ascontiguousarray(counts, dtype=np.int64)
    out = np.empty(sr.shape[0], dtype=np.float64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(lib().fmi_log_odds_batch(sr.shape[0], p(sr), p(cnt), float(ntokens), float(smoothing), p(out)))
    return out


def _unigram_ranges(index):
    """``get_range([i])`` for every token id, one launch, cached on the index."""
    cache = getattr(index, "_unigram_range_cache", None)
    if cache is None:
        vocab = max(index.occurring_distinct) + 1 if index.occurring_distinct else 1
        lo, hi = index.get_range_batch([[t] for t in range(vocab)])
        cache = (np.asarray(lo, dtype=np.int64), np.asarray(hi, dtype=np.int64))
        index._unigram_range_cache = cache
    return cache


def _unigram_counts(index) -> np.ndarray:
    """``get_count([i])`` for every token id (cached)."""
    lo, hi = _unigram_ranges(index)
    return hi - lo


class _Counts:
    """memo of ``index.get_count`` filled by batched launches."""

    def __init__(self, index, shared=None):
        self.index = index
        self.memo: Dict[Tuple[int, ...], int] = {}
        self.ranges: Dict[Tuple[int, ...], Tuple[int, int]] = {}
        self.shared = shared          # ranges prefetched for a whole batch of queries: {tuple: (lo, hi)}

    def ensure(self, ngrams) -> None:
        todo, seen = [], set()
        uni = None
        for ng in ngrams:
            t = tuple(ng)
            if t in self.memo or t in seen:
                continue
            if self.shared is not None and t in self.shared:
                self.ranges[t] = self.shared[t]
                self.memo[t] = self.shared[t][1] - self.shared[t][0]
                continue
            if len(t) == 1:               # single tokens come from the per-index table
                if uni is None:
                    uni = _unigram_ranges(self.index)
                if 0 <= t[0] < len(uni[0]):
                    self.ranges[t] = (int(uni[0][t[0]]), int(uni[1][t[0]]))
                    self.memo[t] = self.ranges[t][1] - self.ranges[t][0]
                    continue
            seen.add(t)
            todo.append(t)
        if not todo:
            return
        lo, hi = self.index.get_range_batch([list(t) for t in todo])
        for t, a, b in zip(todo, lo, hi):
            self.ranges[t] = (int(a), int(b))
            self.memo[t] = int(b) - int(a)

    def __call__(self, ngram) -> int:
        t = tuple(ngram)
        if t not in self.memo:
            self.ensure([t])
        return self.memo[t]


def _match_order_key(length: int) -> Tuple[int, int]:
    # The reference's open-match list is rebuilt by popping from the end on every
    # token (keys.py:400-416); for matches ending on the same token this yields
    # odd lengths in ascending order followed by even lengths in descending order.
    return (0, length) if length % 2 else (1, -length)


class _KeyList:
    """``[[ngram, score], ...]`` of one document, materialised on first use (most of the
    up-to-1500 ranked documents of a query are never looked at again)."""
    __slots__ = ("_keys", "_ki", "_ks", "_a", "_b", "_list")

    def __init__(self, keys, ki, ks, a, b):
        self._keys, self._ki, self._ks, self._a, self._b, self._list = keys, ki, ks, a, b, None

    def _get(self):
        if self._list is None:
            self._list = [[self._keys[int(self._ki[j])], float(self._ks[j])] for j in range(self._a, self._b)]
        return self._list

    def __len__(self):
        return self._b - self._a

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, i):
        return self._get()[i]

    def __eq__(self, other):
        return self._get() == list(other)

    def __repr__(self):
        return repr(self._get())


def _first_stage_native(rare_keys, rare, offs, pos_all, doc_all, allow_overlaps, beta, single_key, n_top, ids_only=False):
    """keys.py:311-367 through ``fmi_first_stage`` (seal_amd/csrc/fmi_evidence.cpp)."""
    import ctypes
    from ._lib import check, lib
    nk = len(rare_keys)
    tok_off = np.zeros(nk + 1, dtype=np.int64)
    if nk:
        np.cumsum([len(k) for k in rare_keys], out=tok_off[1:])
    toks = np.fromiter((t for k in rare_keys for t in k), dtype=np.int64, count=int(tok_off[-1])) if nk else np.zeros(0, np.int64)
    scores = np.asarray([rare[k] for k in rare_keys], dtype=np.float64)
    occ = np.ascontiguousarray(offs, dtype=np.int64)
    pos = np.ascontiguousarray(pos_all, dtype=np.int64)
    doc = np.ascontiguousarray(doc_all, dtype=np.int64)
    ev = ctypes.c_void_p()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(lib().fmi_first_stage(nk, p(tok_off), p(toks), p(scores), p(occ), p(pos), p(doc), int(bool(allow_overlaps)),
                                float(beta), float(single_key), int(n_top), ctypes.byref(ev)))
    try:
        nd, ne = int(lib().fmi_evidence_docs(ev)), int(lib().fmi_evidence_entries(ev))
        d = np.zeros(nd, np.int64); sc = np.zeros(nd, np.float64); bk = np.zeros(nd, np.int64); bs = np.zeros(nd, np.float64)
        ko = np.zeros(nd + 1, np.int64); ki = np.zeros(max(ne, 1), np.int32); ks = np.zeros(max(ne, 1), np.float64)
        check(lib().fmi_evidence_read(ev, p(d), p(sc), p(bk), p(bs), p(ko), p(ki), p(ks)))
    finally:
        lib().fmi_evidence_free(ev)
    if ids_only:                 # the caller goes on to full scoring and only needs the ranked document ids
        return [(x, None) for x in d.tolist()]
    d, sc, bk, bs, ko = d.tolist(), sc.tolist(), bk.tolist(), bs.tolist(), ko.tolist()
    return [(d[i], [sc[i], _KeyList(rare_keys, ki, ks, ko[i], ko[i + 1]), [rare_keys[bk[i]] if bk[i] >= 0 else [], bs[i]]])
            for i in range(nd)]


class _LazyList:
    """a list materialised on first use (documents beyond the caller's top-k are never looked at)"""
    __slots__ = ("_make", "_n", "_list")

    def __init__(self, make, n):
        self._make, self._n, self._list = make, n, None

    def _get(self):
        if self._list is None:
            self._list = self._make()
            self._make = None
        return self._list

    def __len__(self):
        return self._n

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


def _full_score_native(doc_ids, fetched, all_ngrams, unigram_scores, allow_overlaps, beta, single_key,
                       single_key_add_unigrams, unigrams_ignore_free_places):
    """keys.py:377-497 through ``fmi_full_score`` (seal_amd/csrc/fmi_evidence.cpp)."""
    import ctypes
    from ._lib import check, lib
    keys = [k for k, sc in all_ngrams.items() if len(k) >= 1 and sc > 0.0]
    nk = len(keys)
    tok_off = np.zeros(nk + 1, dtype=np.int64)
    if nk:
        np.cumsum([len(k) for k in keys], out=tok_off[1:])
    toks = np.fromiter((t for k in keys for t in k), dtype=np.int64, count=int(tok_off[-1])) if nk else np.zeros(1, np.int64)
    scores = np.asarray([all_ngrams[k] for k in keys], dtype=np.float64) if nk else np.zeros(1, np.float64)
    nd = len(doc_ids)
    if not isinstance(fetched, _DocBatch):
        fetched = _DocBatch(np.concatenate([np.asarray(t, dtype=np.int64) for t in fetched]) if nd else np.zeros(0, np.int64),
                            np.concatenate([[0], np.cumsum([len(t) for t in fetched])]) if nd else np.zeros(1, np.int64))
    # doc_tokens = [2] + get_doc(doc)[:-1] (keys.py:388) for every document at once: shift by one inside each document
    doc_off = fetched.offs
    doc_toks = np.empty(max(len(fetched.flat), 1), dtype=np.int64)
    if len(fetched.flat):
        doc_toks[1:len(fetched.flat)] = fetched.flat[:-1]
        doc_toks[doc_off[:-1][doc_off[:-1] < len(fetched.flat)]] = 2
    doc_arrays = [doc_toks[doc_off[i]:doc_off[i + 1]] for i in range(nd)]
    if unigram_scores is not None:
        ts = np.ascontiguousarray(unigram_scores, dtype=np.float64)
        ts_ptr, vocab = ts.ctypes.data_as(ctypes.c_void_p), ts.shape[0]
    else:
        ts_ptr, vocab = None, 0
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    fs = ctypes.c_void_p()
    check(lib().fmi_full_score(nk, p(tok_off), p(toks), p(scores), ts_ptr, vocab, nd, p(doc_off), p(doc_toks),
                               int(bool(allow_overlaps)), float(beta), float(single_key), int(bool(single_key_add_unigrams)),
                               int(bool(unigrams_ignore_free_places)), ctypes.byref(fs)))
    try:
        n, m = int(lib().fmi_fullscore_docs(fs)), int(lib().fmi_fullscore_entries(fs))
        order = np.zeros(n, np.int64); sc = np.zeros(n, np.float64); bk = np.zeros(n, np.int64); bs = np.zeros(n, np.float64)
        po = np.zeros(n + 1, np.int64); pid = np.zeros(max(m, 1), np.int64); ps = np.zeros(max(m, 1), np.float64)
        check(lib().fmi_fullscore_read(fs, p(order), p(sc), p(bk), p(bs), p(po), p(pid), p(ps)))
    finally:
        lib().fmi_fullscore_free(fs)
    order, sc, bk, bs, po = order.tolist(), sc.tolist(), bk.tolist(), bs.tolist(), po.tolist()

    def picks(a, b):
        return lambda: [((keys[int(pid[j])] if pid[j] >= 0 else (int(-pid[j] - 1),)), float(ps[j])) for j in range(a, b)]

    def tokens(i):
        return lambda: doc_arrays[i].tolist()
    results = {}
    for r in range(n):
        i = order[r]
        results[doc_ids[i]] = [sc[r], _LazyList(picks(po[r], po[r + 1]), po[r + 1] - po[r]), None,
                               _LazyList(tokens(i), len(doc_arrays[i])), [keys[bk[r]] if bk[r] >= 0 else [], bs[r]]]
    return results


def _first_stage_job(payload):
    """picklable unit of work for a host worker process: the native first stage of ONE query,
    returning its top ``keep`` documents with materialised key lists (no GPU involved)."""
    rare_keys, scores, offs, pos_all, doc_all, allow_overlaps, beta, single_key, n_top, keep = payload
    rare = dict(zip(rare_keys, scores))
    ranked = _first_stage_native(rare_keys, rare, offs, pos_all, doc_all, allow_overlaps, beta, single_key,
                                 n_top if keep is None else min(n_top, keep))
    return [(d, [info[0], list(info[1]), info[2]]) for d, info in ranked]


def _full_score_job(payload):
    """picklable unit of work for a host worker process: full-document scoring of ONE query"""
    *args, keep = payload
    res = _full_score_native(*args)
    items = list(res.items())
    if keep is not None:
        items = items[:keep]
    return [(d, [info[0], list(info[1]), None, list(info[3]), info[4]]) for d, info in items]


class _Deferred:
    """result of a first stage running in a worker process; ``result()`` -> {doc: info}"""

    def __init__(self, future):
        self._future = future

    def result(self):
        return dict(self._future.result())


def _first_stage_python(rare_keys, rare, offs, pos_all, doc_all, allow_overlaps, beta, single_key, n_docs_complete_score,
                        sort_by_length, sort_by_freq, count_of):
    """reference keys.py:311-367, python (kept for the sort_by_length / sort_by_freq orders)."""
    def repetition(ngram_set, score, coverage):
        if not coverage:
            return score
        ngram_set = set(ngram_set)
        return (1.0 - beta + (beta * len(ngram_set.difference(coverage)) / len(ngram_set))) * score

    covered = set()
    first_stage: Dict[int, list] = {}
    for ki, ngram in enumerate(rare_keys):
        sco = rare[ngram]
        a, b = int(offs[ki]), int(offs[ki + 1])
        if a == b:
            continue
        pos = pos_all[a:b].astype(np.int64)
        docs = doc_all[a:b].astype(np.int64).tolist()
        m = len(ngram)
        # window of the reference: [tok_end - len, tok_end)  (sic, keys.py:322-323)
        pos_l = pos.tolist()
        sure_new = None
        if len(pos_l) > 1:
            srt = np.sort(pos)
            if m == 0 or bool((np.diff(srt) >= m).all()):
                sure_new = [all((p - j) not in covered for j in range(1, m + 1)) for p in pos_l]
        elif pos_l:
            sure_new = [all((pos_l[0] - j) not in covered for j in range(1, m + 1))]
        done_docs = set()
        for r, (p, doc) in enumerate(zip(pos_l, docs)):
            if sure_new is not None:
                new = sure_new[r]
            else:   # windows of this key overlap each other: sequential, as the reference
                new = all((p - j) not in covered for j in range(1, m + 1))
            info = first_stage.get(doc)
            if info is None:
                info = first_stage[doc] = [0.0, [], [[], 0.0]]
            if sort_by_length:
                better = (m, sco) > (len(info[2][0]), info[2][1])
            elif sort_by_freq:
                better = (-count_of(ngram), sco) > (-count_of(info[2][0]), info[2][1])
            else:
                better = sco > info[2][1]
            if better:
                info[2] = [ngram, sco]
            if new:
                covered.update(range(p - m, p))
            if (new or allow_overlaps) and doc not in done_docs:
                done_docs.add(doc)
                info[0] += sco
                info[1].append((ngram, sco))

    # ---- repetition re-weighting per document (keys.py:352-364) ----
    for info in first_stage.values():
        cover, total = set(), 0.0
        for i, (tt, sco) in enumerate(info[1]):
            tts = set(tt)
            new_sco = repetition(tts, sco, cover)
            total += new_sco
            info[1][i] = [tt, new_sco]
            cover |= tts
        info[0] = total

    return sorted(first_stage.items(),
                    key=lambda kv: (1.0 - single_key) * (-kv[1][0]) + single_key * (-kv[1][2][1]))[:n_docs_complete_score]

def aggregate_evidence(ngrams_and_scores, unigram_scores=None, index=None, max_occurrences_1: int = 1500,
                       max_occurrences_2: int = 10_000_000, n_docs_complete_score: int = 500, alpha: float = 2.0,
                       beta: float = 0.8, length_penalty: float = 0.0, use_fm_index_frequency: bool = True,
                       add_best_unigrams_to_ngrams: bool = False, use_top_k_unigrams=1000, sort_by_length=False,
                       sort_by_freq=False, smoothing=5.0, allow_overlaps=False, single_key=0.0,
                       single_key_add_unigrams=False, unigrams_ignore_free_places=False, first_stage_only=False,
                       defer=None, keep=None):
    """Drop-in for ``seal.keys.aggregate_evidence`` (reference keys.py:178-497).

    ``defer`` (an ``concurrent.futures`` executor) + ``first_stage_only``: the GPU part (counts,
    locate, doc binning) runs here, the host bookkeeping of the first stage is submitted to the
    executor and ``results`` is a handle whose ``result()`` gives the dict; ``keep`` truncates the
    ranking to its first ``keep`` documents.

    Returns ``(results, all_ngrams)``: ``results`` maps doc index ->
    ``[score, [(ngram, score)...], None, doc_tokens, [best_ngram, best_score]]``
    sorted by descending score.  ``first_stage_only=True`` (an addition) stops
    after the first stage + repetition re-weighting and returns the
    ``to_fully_score`` ranking as ``{doc: [score, [[ngram, score]...], [best_ngram, best_score]]}``.
    """
    params = dict(max_occurrences_1=max_occurrences_1, max_occurrences_2=max_occurrences_2, n_docs_complete_score=n_docs_complete_score,
                  alpha=alpha, beta=beta, length_penalty=length_penalty, use_fm_index_frequency=use_fm_index_frequency,
                  add_best_unigrams_to_ngrams=add_best_unigrams_to_ngrams, use_top_k_unigrams=use_top_k_unigrams,
                  sort_by_length=sort_by_length, sort_by_freq=sort_by_freq, smoothing=smoothing, allow_overlaps=allow_overlaps,
                  single_key=single_key, single_key_add_unigrams=single_key_add_unigrams,
                  unigrams_ignore_free_places=unigrams_ignore_free_places, first_stage_only=first_stage_only, defer=defer, keep=keep)
    if defer is None and gpu_aggregation_applies(index, params):
        # first stage + full-document scoring on the GPU (one-query chunk of aggregate_evidence_batch)
        return aggregate_evidence_batch([(ngrams_and_scores, unigram_scores)], index, **params)[0]
    gen = _aggregate_steps(ngrams_and_scores, unigram_scores, index, max_occurrences_1, max_occurrences_2,
                           n_docs_complete_score, alpha, beta, length_penalty, use_fm_index_frequency,
                           add_best_unigrams_to_ngrams, use_top_k_unigrams, sort_by_length, sort_by_freq, smoothing,
                           allow_overlaps, single_key, single_key_add_unigrams, unigrams_ignore_free_places,
                           first_stage_only, defer, keep)
    try:
        req = next(gen)
        while True:
            if req[0] == "locate":
                req = gen.send(index.locate_ranges(req[1], req[2], req[3]))
            else:
                req = gen.send(_fetch_docs(index, req[1]))
    except StopIteration as done:
        return done.value


class _DocBatch:
    """documents of a request as one flat int64 array + offsets (cheap to slice, cheap to pickle)"""

    def __init__(self, flat, offs):
        self.flat, self.offs = np.ascontiguousarray(flat, dtype=np.int64), np.ascontiguousarray(offs, dtype=np.int64)

    def __len__(self):
        return len(self.offs) - 1

    def __iter__(self):
        return (self.flat[self.offs[i]:self.offs[i + 1]] for i in range(len(self)))

    def slice(self, a, b):
        o = self.offs[a:b + 1]
        return _DocBatch(self.flat[o[0]:o[-1]], o - o[0])


def _fetch_docs(index, doc_ids):
    try:
        flat, offs = index.get_docs_batch(doc_ids, as_arrays="flat")
        return _DocBatch(flat, offs)
    except TypeError:
        docs = index.get_docs_batch(doc_ids)
        offs = np.zeros(len(docs) + 1, dtype=np.int64)
        if docs:
            np.cumsum([len(d) for d in docs], out=offs[1:])
        flat = np.fromiter((t for d in docs for t in d), dtype=np.int64, count=int(offs[-1])) if docs else np.zeros(0, np.int64)
        return _DocBatch(flat, offs)


def gpu_aggregation_applies(index, params) -> bool:
    from .gpu_aggregate import gpu_aggregation_applies as applies
    return applies(index, params)


def _aggregate_on_gpu(index, requests, params):
    from .gpu_aggregate import aggregate_on_gpu
    return aggregate_on_gpu(index, requests, params)


def aggregate_evidence_batch(jobs, index, **params):
    """``aggregate_evidence`` for several queries with the GPU work batched ACROSS them: one
    backward-search launch for every key of every query, one locate launch for every rare key of
    every query (and one document fetch when fully scoring).  ``jobs`` = list of
    ``(ngrams_and_scores, unigram_scores)``; returns the list of ``(results, all_ngrams)``.

    ``two_phase=True``: returns a callable that returns that list.  On the GPU route the call itself scores the keys and ENQUEUES the index
    kernels, and the callable waits for them and builds the results (``score_and_aggregate_on_gpu``); everywhere else the callable does
    all the work."""
    import os, time, sys
    if params.pop("two_phase", False):
        use_gpu = params.get("gpu_aggregate", True)
        if (use_gpu and not params.get("python_scoring", False)
                and gpu_aggregation_applies(index, {k: v for k, v in params.items() if k not in ("gpu_aggregate", "want_ngrams", "python_scoring")})):
            from .gpu_aggregate import score_and_aggregate_on_gpu
            rest = {k: v for k, v in params.items() if k not in ("gpu_aggregate", "want_ngrams", "python_scoring")}
            fetch = score_and_aggregate_on_gpu(index, jobs, rest, params.get("want_ngrams", True), two_phase=True)

            def finish():
                done = fetch()
                if done is not None and all(r is not None for r in done):
                    return done
                if done is None:
                    return aggregate_evidence_batch(jobs, index, **{**params, "python_scoring": True})
                left = [i for i, r in enumerate(done) if r is None]
                some = aggregate_evidence_batch([jobs[i] for i in left], index, **{**params, "gpu_aggregate": False})
                for i, r in zip(left, some):
                    done[i] = r
                return done
            return finish
        return lambda: aggregate_evidence_batch(jobs, index, **params)
    use_gpu = params.pop("gpu_aggregate", True)          # False: the host routines (fmi_first_stage / fmi_full_score) for every query
    want_ngrams = params.pop("want_ngrams", True)        # False: the caller ignores `all_ngrams` (the searcher does)
    python_scoring = params.pop("python_scoring", False)
    _tm = AGG_TIMING
    _t = {"t0": time.perf_counter()}
    def _mark(name):
        if _tm:
            now = time.perf_counter(); _t[name] = _t.get(name, 0.0) + (now - _t["t0"]) * 1e3; _t["t0"] = now
    done_on_gpu = None
    if use_gpu and not python_scoring and gpu_aggregation_applies(index, params):
        # key scoring in C++ (fmi_agg_score_pack), everything per row / per document on the GPU (fmi_dev_aggregate);
        # the python below (generators, python scoring) stays for what that route hands back
        from .gpu_aggregate import score_and_aggregate_on_gpu
        done_on_gpu = score_and_aggregate_on_gpu(index, jobs, params, want_ngrams)
        _mark("score_pack_gpu")
        if done_on_gpu is not None and all(r is not None for r in done_on_gpu):
            if _tm:
                _t.pop("t0")
                print("[agg]", {k: round(v, 1) for k, v in _t.items()}, file=sys.stderr, flush=True)
            return done_on_gpu
        if done_on_gpu is not None:
            left = [i for i, r in enumerate(done_on_gpu) if r is None]
            rest = aggregate_evidence_batch([jobs[i] for i in left], index, gpu_aggregate=False, want_ngrams=want_ngrams, **params)
            for i, r in zip(left, rest):
                done_on_gpu[i] = r
            return done_on_gpu
    all_keys, seen = [], set()
    for ngrams_and_scores, _ in jobs:
        for ng, _ in ngrams_and_scores:
            t = tuple(ng.tolist() if isinstance(ng, torch.Tensor) else ng)
            if t not in seen:
                seen.add(t)
                all_keys.append(t)
    shared = {}
    if all_keys:
        lo, hi = index.get_range_batch([list(t) for t in all_keys])
        shared = {t: (int(a), int(b)) for t, a, b in zip(all_keys, lo, hi)}
    _mark("ranges")
    gens = [_aggregate_steps(nas, us, index, shared_ranges=shared, **params) for nas, us in jobs]
    out = [None] * len(gens)
    reqs = {}
    for i, g in enumerate(gens):
        try:
            reqs[i] = next(g)
        except StopIteration as done:
            out[i] = done.value
    _mark("score_split")
    if reqs and use_gpu and gpu_aggregation_applies(index, params):
        # first stage + full-document scoring on the GPU (seal_amd/csrc/fmi_aggregate.hip): nothing but the top
        # documents comes back.  A query that exceeds a device limit is handed back to the host routines below.
        live = sorted(reqs)
        results = _aggregate_on_gpu(index, [reqs[i] for i in live], params)
        _mark("gpu_aggregate")
        for i, res in zip(live, results):
            if res is not None:
                out[i] = (res, reqs[i][4]["all_ngrams"])
                gens[i].close()
                del reqs[i]
    while reqs:
        loc = [i for i, r in reqs.items() if r[0] == "locate"]
        answers = {}
        if loc:
            mx = reqs[loc[0]][3]
            los = np.concatenate([np.asarray(reqs[i][1], dtype=np.int64) for i in loc])
            his = np.concatenate([np.asarray(reqs[i][2], dtype=np.int64) for i in loc])
            pos, doc, offs = index.locate_ranges(los, his, mx)
            k0 = 0
            for i in loc:
                nk = len(reqs[i][1])
                o = offs[k0:k0 + nk + 1]
                answers[i] = (pos[o[0]:o[-1]], doc[o[0]:o[-1]], o - o[0])
                k0 += nk
            _mark("locate")
        dcs = [i for i, r in reqs.items() if r[0] == "docs"]
        if dcs:
            flat = [d for i in dcs for d in reqs[i][1]]
            fetched = _fetch_docs(index, flat)
            k0 = 0
            for i in dcs:
                nd = len(reqs[i][1])
                answers[i] = fetched.slice(k0, k0 + nd)
                k0 += nd
            _mark("fetch_docs")
        nxt = {}
        for i, ans in answers.items():
            try:
                nxt[i] = gens[i].send(ans)
            except StopIteration as done:
                out[i] = done.value
        _mark("host_after_docs" if dcs else "host_after_locate")
        reqs = nxt
    if _tm:
        _t.pop("t0")
        print("[agg]", {k: round(v, 1) for k, v in _t.items()}, file=sys.stderr, flush=True)
    return out


def _aggregate_steps(ngrams_and_scores, unigram_scores=None, index=None, max_occurrences_1: int = 1500,
                       max_occurrences_2: int = 10_000_000, n_docs_complete_score: int = 500, alpha: float = 2.0,
                       beta: float = 0.8, length_penalty: float = 0.0, use_fm_index_frequency: bool = True,
                       add_best_unigrams_to_ngrams: bool = False, use_top_k_unigrams=1000, sort_by_length=False,
                       sort_by_freq=False, smoothing=5.0, allow_overlaps=False, single_key=0.0,
                       single_key_add_unigrams=False, unigrams_ignore_free_places=False, first_stage_only=False,
                       defer=None, keep=None, shared_ranges=None):
    """generator form of ``aggregate_evidence``: yields ("locate", los, his, max) / ("docs", ids) requests
    that a driver may batch across queries, receives their answers through ``send`` and returns the result."""
    def repetition(ngram_set, score, coverage):
        if not coverage:
            return score
        ngram_set = set(ngram_set)
        return (1.0 - beta + (beta * len(ngram_set.difference(coverage)) / len(ngram_set))) * score

    ntokens = float(index.beginnings[-1])
    keys: List[Tuple[List[int], float]] = [
        (ng.tolist() if isinstance(ng, torch.Tensor) else list(ng), sr) for ng, sr in ngrams_and_scores]
    count_of = _Counts(index, shared_ranges)
    count_of.memo[tuple()] = len(index)
    count_of.ensure([ng for ng, _ in keys])
    cutoff = None
    if not use_fm_index_frequency:
        cutoff = min(s for _, s in keys) - 0.1 if keys else None
        if cutoff is None:
            raise IndexError("list index out of range")

    # ---- key scores (keys.py:207-234), vectorised; the log-odds go through libm in C exactly as
    #      python's math module would compute them one key at a time ----
    seen_unigrams = {0, 1, 2}
    for ng, _ in keys:
        if len(ng) == 1:
            seen_unigrams.add(ng[0])
    scored: List[Tuple[List[int], float]] = []
    if keys:
        cnts = np.fromiter((count_of(ng) for ng, _ in keys), dtype=np.int64, count=len(keys))
        # powers stay python floats (C pow, as the reference's `**`): numpy's vectorised pow may differ by an ulp
        lp_factor = [(1.0 - length_penalty) ** (len(ng) - 1.0) for ng, _ in keys]
        if use_fm_index_frequency:
            sr_adj = np.asarray([(sr - 1e-10) * f for (_, sr), f in zip(keys, lp_factor)], dtype=np.float64)
            odds = _log_odds_many(sr_adj, cnts, ntokens, smoothing).tolist()
            sco = [max(o, 0.0) ** alpha for o in odds]
        else:
            sco = [(max(sr - cutoff, 0.0) * f) ** alpha for (_, sr), f in zip(keys, lp_factor)]
        sco = [0.0 if c == 0 else v for c, v in zip(cnts.tolist(), sco)]
        scored = [(ng, float(v)) for (ng, _), v in zip(keys, sco)]

    # ---- unigram scores (keys.py:236-278) ----
    if unigram_scores is not None:
        raw = np.asarray(unigram_scores, dtype=np.float64)
        V = raw.shape[0]
        # the use_top_k_unigrams best by log-prob, ties to the lower id (the reference's
        # sorted(range(V), reverse=True, key=...) is stable) -- selected without a full sort
        k = min(int(use_top_k_unigrams), V)
        if k <= 0:
            best = np.zeros(0, dtype=np.int64)
        elif k >= V:
            best = np.arange(V)
        else:
            kth = np.partition(raw, V - k)[V - k]
            above = np.flatnonzero(raw > kth)
            ties = np.flatnonzero(raw == kth)[:k - above.size]
            best = np.concatenate([above, ties])
        uni_counts = _unigram_counts(index)
        us = np.zeros(V, dtype=np.float64)
        cand = best[~np.isin(best, np.fromiter(seen_unigrams, dtype=np.int64))]
        cand = cand[cand < len(uni_counts)]
        cand = cand[uni_counts[cand] > 0]
        if cand.size:
            if use_fm_index_frequency:
                sco = np.maximum(_log_odds_many(raw[cand], uni_counts[cand], ntokens, smoothing), 0.0)
            else:
                sco = np.asarray([max(v - cutoff, 0.0) ** alpha for v in raw[cand].tolist()], dtype=np.float64)
            us[cand] = sco
        unigram_scores = us          # indexable by token id like the reference's list
        if add_best_unigrams_to_ngrams:
            # sorted(range(V), key=lambda x: -unigram_scores[x])[:n]: positives by descending score
            # (stable), then the zero-score ids in ascending order
            n_add = len(scored)
            nz = np.flatnonzero(us > 0)
            order = nz[np.argsort(-us[nz], kind="stable")][:n_add].tolist()
            if len(order) < n_add:
                taken = set(order)
                for i in range(V):
                    if len(order) >= n_add:
                        break
                    if us[i] == 0.0 and i not in taken:
                        order.append(i)
            for i in order:
                scored.append(([i], float(us[i])))
        count_of.ensure([ng for ng, _ in scored])

    # ---- rare / frequent split (keys.py:280-309) ----
    rare: Dict[Tuple[int, ...], float] = {}
    freq: Dict[Tuple[int, ...], float] = {}
    for ng, sco in scored:
        count = count_of(ng)
        if count > max_occurrences_2 or sco == 0.0:
            continue
        (freq if (count > max_occurrences_1 or sco < 0.0) else rare)[tuple(ng)] = sco
    rare = dict(sorted(rare.items(), key=lambda kv: kv[1], reverse=True))
    freq = dict(sorted(freq.items(), key=lambda kv: kv[1], reverse=True))
    all_ngrams = dict(sorted(list(rare.items()) + list(freq.items()), key=lambda kv: kv[1], reverse=True))

    # ---- first stage: locate + doc binning for every row of every rare key,
    #      one launch (keys.py:311-350) ----
    rare_keys = list(rare.keys())
    count_of.ensure(rare_keys)
    if rare_keys:
        los = np.asarray([count_of.ranges[k][0] if k in count_of.ranges else 0 for k in rare_keys], dtype=np.uint64)
        his = np.asarray([count_of.ranges[k][1] if k in count_of.ranges else 0 for k in rare_keys], dtype=np.uint64)
        # the 5th element hands the scored keys to a driver that runs the rest on the GPU (_aggregate_on_gpu)
        pos_all, doc_all, offs = yield ("locate", los, his, max_occurrences_1,
                                        dict(rare=rare, all_ngrams=all_ngrams, unigram_scores=unigram_scores))
    else:
        pos_all = doc_all = np.zeros(0, dtype=np.int64)
        offs = np.zeros(1, dtype=np.int64)

    if defer is not None and first_stage_only and not (sort_by_length or sort_by_freq):
        payload = (rare_keys, [rare[k] for k in rare_keys], np.asarray(offs), np.asarray(pos_all), np.asarray(doc_all),
                   allow_overlaps, beta, single_key, n_docs_complete_score, keep)
        return _Deferred(defer.submit(_first_stage_job, payload)), all_ngrams
    if not (sort_by_length or sort_by_freq):
        # native host routine (libsealfm fmi_first_stage): same bookkeeping, same float64
        # operation order, ~100x faster than the python loop below
        ranked = _first_stage_native(rare_keys, rare, offs, pos_all, doc_all, allow_overlaps, beta, single_key,
                                     n_docs_complete_score, ids_only=not first_stage_only)
    else:
        ranked = _first_stage_python(rare_keys, rare, offs, pos_all, doc_all, allow_overlaps, beta, single_key,
                                     n_docs_complete_score, sort_by_length, sort_by_freq, count_of)
    if first_stage_only:
        return dict(ranked[:keep] if keep is not None else ranked), all_ngrams

    # ---- full scoring of the top documents (keys.py:366-497) ----
    doc_ids = [d for d, _ in ranked]
    fetched = (yield ("docs", doc_ids)) if doc_ids else _DocBatch(np.zeros(0, np.int64), np.zeros(1, np.int64))
    if defer is not None and not (sort_by_length or sort_by_freq):
        payload = (doc_ids, fetched, all_ngrams,
                   None if unigram_scores is None else np.asarray(unigram_scores, dtype=np.float64), allow_overlaps, beta,
                   single_key, single_key_add_unigrams, unigrams_ignore_free_places, keep)
        return _Deferred(defer.submit(_full_score_job, payload)), all_ngrams
    if not (sort_by_length or sort_by_freq):
        # native host routine (libsealfm fmi_full_score): trie matching, the reference's registration
        # and heap orders, greedy non-overlapping selection, unigram fill -- float64, same operation order
        return _full_score_native(doc_ids, fetched, all_ngrams, unigram_scores, allow_overlaps, beta, single_key,
                                  single_key_add_unigrams, unigrams_ignore_free_places), all_ngrams
    # python form (kept for the sort_by_length / sort_by_freq orders)
    trie: dict = {}
    for ngram, score in all_ngrams.items():
        if len(ngram) < 1 or score <= 0.0:
            continue
        node = trie
        for t in ngram:
            node = node.setdefault(t, {})
        node[-1] = score
    results: Dict[int, list] = {}
    for doc, toks in zip(doc_ids, fetched):
        doc_tokens = [2] + [int(x) for x in toks][:-1]
        res = results[doc] = [0.0, [], None, doc_tokens, [[], 0.0]]
        if unigram_scores is not None:
            type_scores = {t: float(unigram_scores[t]) for t in doc_tokens}
        else:
            type_scores = {t: 0.0 for t in doc_tokens}
        # all trie matches, bucketed by end position
        by_end: List[list] = [[] for _ in doc_tokens]
        T = len(doc_tokens)
        for s in range(T):
            node = trie.get(doc_tokens[s])
            e = s
            while node is not None:
                if -1 in node:
                    by_end[e].append((e - s + 1, s, node[-1]))
                e += 1
                if e >= T:
                    break
                node = node.get(doc_tokens[e])
        matches: Dict[Tuple[int, ...], list] = {}
        for e in range(T):
            for length, s, score in sorted(by_end[e], key=lambda x: _match_order_key(x[0])):
                key = tuple(doc_tokens[s:e + 1])
                matches.setdefault(key, [score, []])[1].append((s, e + 1))
        cand = []
        for n, (s, spans) in matches.items():
            if sort_by_length:
                better = (-len(n), -s) < (-len(res[4][0]), -res[4][1])
            elif sort_by_freq:
                better = (count_of(n), -s) < (count_of(res[4][0]), -res[4][1])
            else:
                better = -s < -res[4][1]
            for (i, j) in spans:
                cand.append((-s, n, s, i, j))
            if better:
                res[4] = [n, s]
        cand.sort()     # == successive heappop of the reference's heap
        cover = set()
        picked: List[Tuple[Tuple[int, ...], float]] = []
        prev = None
        free = [True] * T
        for _, n, s, i, j in cand:
            n_set = set(n)
            if prev == n:
                new_s = picked[-1][1]
            elif not n_set:
                new_s = 0.0
            else:
                new_s = repetition(n_set, s, cover)
            if new_s <= 0.0:
                continue
            if not (allow_overlaps or all(free[i:j])):
                continue
            if prev == n:
                picked[-1] = (n, new_s)
            else:
                prev = n
                cover |= n_set
                picked.append((n, new_s))
            free[i:j] = [False] * (j - i)
        if unigrams_ignore_free_places:
            free = [True] * T
        single_key_score = res[4][1]
        multi_key_score = sum(s for _, s in picked)
        unigram_score = 0.0
        for t in Counter(t for t, f in zip(doc_tokens, free) if f):
            s = type_scores[t]
            if s > 0.0:
                s = repetition((t,), s, cover)
                if s != 0.0:
                    unigram_score += s
                    picked.append(((t,), s))
        if single_key_add_unigrams:
            single_key_score += unigram_score
        multi_key_score += unigram_score
        res[0] = (1.0 - single_key) * multi_key_score + single_key * single_key_score
        res[1] = picked
    results = dict(sorted(results.items(), key=lambda kv: -kv[1][0]))
    return results, all_ngrams
