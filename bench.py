#!/usr/bin/env python
"""bench.py -- queries/sec of the SEAL search hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic queries:
the reference's complete ``SEALSearcher.batch_search`` (body decode 10 tokens +
title decode <= 15 tokens, beam 15, FM-index constrained; count post-filters;
rescoring; unigram scores; first-stage retrieval = counts + locate + doc binning +
evidence aggregation; full-document rescoring of the 1500 best documents per query,
keys.py:366-497 -- both aggregation stages as GPU kernels) on an NQ-shaped synthetic
FM-index with a random-init BART-large (fp32, as the reference runs it); the query
n-gram keys of the reference's default configuration (add_query_to_keys) are in
the step in their token-id form.  After the
timed region the GPU's answers for one batch are compared with the CPU oracle's
and with the bit-exact host routines (``parity_check`` in the JSON line; a mismatch
exits non-zero).

Contract: python bench.py --gpus N --steps K --warmup W ; for N > 1 launched by
torch.distributed.run, one rank per GPU; prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHIFT = 10
TITLE_EOS, CODE_EOS, VOCAB = 49314, 45056, 50265
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------
# synthetic NQ-shaped corpus, generated on the GPU directly in
# index order: per document reversed (+SHIFT), i.e. [</s>, body..., '@@', title...]
# ---------------------------------------------------------------------------
def _sym(v: int, dtype):
    """symbol value as `dtype` stores it (int16 = the two's-complement view of the 16-bit symbol)"""
    return v - 65536 if dtype == torch.int16 and v >= 32768 else v


def synth_corpus(n_docs: int, device, seed: int = 0, phrases: int = 0, text16: bool = False):
    """``text16``: the corpus as the index's resident text itself -- int16 (two's-complement view of the 16-bit symbols), one element
    longer, ending with the 0 sentinel -- for the tier whose int32 form (56 GB at 1.4e10 symbols) has no room next to the build.
    ``phrases`` = P > 0 (bench.py's default, P = 20 M): the token stream is a concatenation of phrases of 2..8 tokens
    drawn Zipf(1.0) from a dictionary of P phrases (each an i.i.d. Zipf(1.07) token string), cut into documents
    independently of the phrase boundaries: n-grams repeat as they do in real text, so a key locates ~3e5 rows per query
    (a real NQ index: 1e5-1e6) instead of the ~1e4 of an i.i.d. corpus.
    ``phrases`` = 0: every token drawn i.i.d. Zipf(1.07) (round 1's workload; tools/expand_bench.py keeps it)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = torch.normal(137.0, 25.0, (n_docs,), generator=g, device=device).round().clamp(40, 256).long()
    title_len = torch.randint(2, 9, (n_docs,), generator=g, device=device)
    beg = torch.zeros(n_docs + 1, dtype=torch.long, device=device)
    torch.cumsum(lens, 0, out=beg[1:])
    N = int(beg[-1])
    usable = torch.arange(4, VOCAB, device=device)
    usable = usable[(usable != TITLE_EOS) & (usable != CODE_EOS)]
    ids_by_rank = usable[torch.randperm(usable.numel(), generator=g, device=device)]
    w = 1.0 / torch.arange(1, usable.numel() + 1, device=device, dtype=torch.float64) ** 1.07
    cdf = torch.cumsum(w, 0) / w.sum()
    dt = torch.int16 if text16 else torch.int32
    data = torch.empty(N + (1 if text16 else 0), dtype=dt, device=device)
    if text16:
        data[N] = 0
    CH = 1 << 27

    def zipf_tokens(n):
        u = torch.rand(n, generator=g, device=device, dtype=torch.float64)
        r = torch.searchsorted(cdf, u).clamp_(max=usable.numel() - 1)
        return (ids_by_rank[r] + SHIFT).to(torch.int32)
    if phrases <= 0:
        for a in range(0, N, CH):
            b = min(N, a + CH)
            data[a:b] = zipf_tokens(b - a)
    else:
        LMAX = 8
        phrases = max(16, min(phrases, N // 100))      # a small test corpus gets a proportionally small dictionary
        ptab = zipf_tokens(phrases * LMAX).view(phrases, LMAX)
        plen = torch.randint(2, LMAX + 1, (phrases,), generator=g, device=device)
        pw = 1.0 / torch.arange(1, phrases + 1, device=device, dtype=torch.float64)
        pcdf = torch.cumsum(pw, 0) / pw.sum()
        a = 0
        while a < N:                                   # chunks of phrase slots; a chunk's tail phrase is cut at the chunk end
            b = min(N, a + CH)
            n_slots = (b - a) // 2 + 1                # enough slots even if every phrase had the minimum length
            u = torch.rand(n_slots, generator=g, device=device, dtype=torch.float64)
            pid = torch.searchsorted(pcdf, u).clamp_(max=phrases - 1)
            cum = torch.cumsum(plen[pid], 0)
            pos = torch.arange(b - a, device=device)
            slot = torch.searchsorted(cum, pos, right=True)
            start = torch.where(slot > 0, cum[(slot - 1).clamp_(min=0)], torch.zeros_like(pos))
            data[a:b] = ptab[pid[slot], pos - start]
            del u, pid, cum, pos, slot, start
            a = b
    data[beg[:-1]] = _sym(2 + SHIFT, dt)
    data[beg[1:] - 1 - title_len] = _sym(TITLE_EOS + SHIFT, dt)
    return data, beg, title_len, ids_by_rank


def synth_queries(n, data, beg, title_len, ids_by_rank, device, seed: int = 1):
    """encoder token ids [<s>, 8..24 tokens, </s>] + per-query logit bias (+8) on the tokens of a
    corpus 10-gram and of the same document's title, so that a random-init model walks real corpus
    paths with NQ-like interval sizes."""
    rng = np.random.default_rng(seed)
    n_docs = beg.numel() - 1
    queries, bias = [], torch.zeros(n, VOCAB, device=device)
    V2 = ids_by_rank.numel()
    for q in range(n):
        m = int(rng.integers(8, 25))
        ranks = np.minimum(rng.zipf(1.3, size=m), V2) - 1
        queries.append([0] + ids_by_rank[torch.as_tensor(ranks, device=device)].tolist() + [2])
        d = int(rng.integers(0, n_docs))
        b, e, tl = int(beg[d]), int(beg[d + 1]), int(title_len[d])
        rev = ((data[b:e].long() & 0xFFFFF) - SHIFT) if data.dtype != torch.int16 else ((data[b:e].long() & 0xFFFF) - SHIFT)
        fwd = torch.flip(rev, [0])                       # title..., '@@', body..., </s>
        body0 = tl + 1
        s = int(rng.integers(body0, max(body0 + 1, (e - b) - 11)))
        toks = torch.cat([fwd[s:s + 10], fwd[:tl]])
        bias[q, toks] = 8.0
    return queries, bias


class _CudaArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_array(index, name, typestr):
    import ctypes
    from seal_amd._lib import lib
    n, e = ctypes.c_uint64(), ctypes.c_uint32()
    p = lib().fmi_dev_array(index.handle, name.encode(), ctypes.byref(n), ctypes.byref(e))
    if not p:
        return None
    return torch.as_tensor(_CudaArray(p, n.value, typestr), device=torch.device("cuda", torch.cuda.current_device()))


# ---------------------------------------------------------------------------
# CPU baseline: the oracle (oracle/, a restatement of the sdsl path), replaying the
# FM-index operations of one batch with the reference's call pattern.
# ---------------------------------------------------------------------------
def build_cpu_oracle(index, threads):
    from oracle.seal_oracle import lib as orc_lib
    orc_lib().orc_set_threads(threads)
    n = index.size()
    sa = device_array(index, "sa_lo", "<i4")
    sa_hi = device_array(index, "sa_hi", "|u1")           # bits 32..39 of the suffix array above 2^32 rows (KILT size)
    text = device_array(index, "text", "<i2" if index_sym_bytes(index) == 2 else "<i4")
    bwt = np.empty(n, dtype=np.uint32)
    isa_s = torch.zeros(n // 64 + 1, dtype=torch.int64, device=sa.device)
    CH = 1 << 27
    for a in range(0, n, CH):
        b = min(n, a + CH)
        pos = sa[a:b].long() & 0xFFFFFFFF
        if sa_hi is not None:
            pos |= sa_hi[a:b].long() << 32
        prev = torch.where(pos == 0, torch.full_like(pos, n - 1), pos - 1)
        sym = text[prev].to(torch.int32) & (0xFFFF if text.dtype == torch.int16 else 0x7FFFFFFF)
        bwt[a:b] = sym.cpu().numpy().astype(np.uint32)
        m = (pos & 63) == 0
        isa_s[pos[m] >> 6] = torch.arange(a, b, device=sa.device)[m]
        del pos, prev, sym, m
    sa_s = sa[::32].long() & 0xFFFFFFFF
    if sa_hi is not None:
        sa_s |= sa_hi[::32].long() << 32
    sa_s = sa_s.cpu().numpy().astype(np.uint64)
    orc = _scalar_oracle_class()()
    orc.initialize_from_bwt(bwt, sa_s, isa_s.cpu().numpy().astype(np.uint64))
    orc.beginnings = index.beginnings
    orc.batch_threads = threads
    return orc


def _scalar_oracle_class():
    from oracle.seal_oracle import OracleFMIndex

    class ScalarOracleIndex(OracleFMIndex):
        """The reference's ``seal.index.FMIndex`` interface (index.py:68-118, restated by ``OracleFMIndex``) over the NQ-scale oracle index:
        what oracle/keys_oracle.py queries one scalar call at a time.  ``locate`` answers from the oracle's own batched entry point
        (the same LF walks, on all host threads): a miss locates the rest of the current key's rows in one call."""
        batch_threads = 1

        def __init__(self):
            super().__init__()
            self._located, self._range_end = {}, 0
            self.seconds = {"locate_batches": 0.0, "get_doc": 0.0}

        def get_doc(self, doc_index):
            t0 = time.perf_counter()
            out = super().get_doc(doc_index)
            self.seconds["get_doc"] += time.perf_counter() - t0
            return out

        def get_range(self, sequence):
            lo, hi = super().get_range(sequence)
            self._range_end = hi
            return lo, hi

        def locate(self, row):
            pos = self._located.get(row)
            if pos is None:
                t0 = time.perf_counter()
                rows = np.arange(row, max(row + 1, min(self._range_end, row + 1500)), dtype=np.uint64)
                got, _ = self.locate_bin_batch(rows, np.asarray([0], dtype=np.uint64), threads=min(16, self.batch_threads))
                self.seconds["locate_batches"] += time.perf_counter() - t0
                self._located.update(zip(rows.tolist(), got.astype(np.int64).tolist()))
                pos = self._located[row]
            return pos
    return ScalarOracleIndex


def aggregation_vs_keys_oracle(orc, agg_calls, n_queries=2):
    """The evidence aggregation of the timed path's kernels (fmi_aggregate.hip) against oracle/keys_oracle.py -- the independent scalar
    model of the reference's ``aggregate_evidence`` (keys.py:178-497) that tests/test_reference_golden.py pins to the reference's own
    outputs -- on a SAMPLE of the recorded batch's queries, at this index's real size: ranked documents, float64 scores (bit for bit),
    accepted keys with their discounted scores, document tokens, best key.  The searcher's parameters are passed through as recorded."""
    from oracle.keys_oracle import oracle_aggregate_evidence
    names = ("max_occurrences_1", "max_occurrences_2", "n_docs_complete_score", "alpha", "beta", "length_penalty", "use_fm_index_frequency",
             "add_best_unigrams_to_ngrams", "use_top_k_unigrams", "sort_by_length", "sort_by_freq", "smoothing", "allow_overlaps", "single_key",
             "single_key_add_unigrams", "unigrams_ignore_free_places", "first_stage_only")
    n_docs = n_bad = n_q = located = 0
    first_bad = None
    for a, kw, res in agg_calls:
        jobs = a[0]
        okw = {k: kw[k] for k in names if k in kw}
        for (ngrams, uni), (got, _) in list(zip(jobs, res))[:max(0, n_queries - n_q)]:
            got = got.result() if hasattr(got, "result") else got
            keys = [([int(t) for t in (ng.tolist() if hasattr(ng, "tolist") else ng)], float(sc)) for ng, sc in ngrams]
            us = None if uni is None else [float(x) for x in (uni.tolist() if hasattr(uni, "tolist") else uni)]
            orc._located.clear()
            want, _ = oracle_aggregate_evidence(keys, unigram_scores=us, index=orc, **okw)
            located += len(orc._located)
            wl = list(want.items())[:len(got)]
            n_q += 1
            n_docs += len(wl)
            n_bad += abs(len(got) - len(wl))
            for (gd, gi), (wd, wi) in zip(got.items(), wl):
                same = (int(gd) == int(wd) and float(gi[0]).hex() == float(wi[0]).hex()
                        and [(tuple(int(t) for t in k), float(v).hex()) for k, v in gi[1]] == [(tuple(int(t) for t in k), float(v).hex()) for k, v in wi[1]]
                        and [int(t) for t in gi[3]] == [int(t) for t in wi[3]]
                        and tuple(int(t) for t in gi[4][0]) == tuple(int(t) for t in wi[4][0]) and float(gi[4][1]).hex() == float(wi[4][1]).hex())
                if not same:
                    n_bad += 1
                    first_bad = first_bad or {"query": n_q - 1, "gpu_doc": int(gd), "oracle_doc": int(wd), "gpu_score": float(gi[0]).hex(), "oracle_score": float(wi[0]).hex()}
    out = {"ops": n_q, "values": n_docs, "mismatches": n_bad, "queries": n_q, "rows_located_by_the_oracle": located,
           "oracle_seconds": {k: round(v, 1) for k, v in orc.seconds.items()},
           "against": "oracle/keys_oracle.py (scalar model of keys.py:178-497, pinned to the reference's own outputs) over the sdsl-style oracle index of this "
                      "corpus: document order, float64 scores bit for bit, accepted keys + discounted scores, document tokens, best key"}
    if first_bad:
        out["first_mismatch"] = first_bad
    return out


def index_sym_bytes(index):
    from seal_amd._lib import lib
    return 2 if lib().fmi_max_symbol(index.handle) < 65536 else 4


def replay_on_cpu(orc, trace, beginnings, threads, vocab=VOCAB, locate_stride=1):
    """reference call pattern: per decode step and row, get_range(prefix) and
    get_count(prefix[:-1]) from scratch (beam_search.py:96-101), one task per row for
    distinct_count_multi (fm_index.cpp:117-121); get_count per key; locate + bisect per row;
    extract_text per document.  Returns the timings AND every answer (for the parity check)."""
    t_mask = t_rng = t_loc = t_doc = 0.0
    n_rows = n_seq = n_loc = n_doc = 0
    b = np.asarray(beginnings, dtype=np.uint64)
    answers = []
    for op in trace:
        if op[0] == "mask":
            ids, ff, kw = op[1].tolist(), op[2], op[3]
            seqs, live = [], []
            for r, sent in enumerate(ids):
                if sent[-1] in (kw["eos"], kw["pad"]):
                    continue
                live.append(r)
                seqs.append(ff + sent[1:])
                seqs.append(ff + sent[1:-1])
            t0 = time.perf_counter()
            lo, hi = orc.get_range_batch(seqs, threads=threads)
            bits, k, cs = orc.distinct_bitmaps(lo[0::2], np.maximum(lo[0::2], np.minimum(hi[0::2], orc.size())), vocab, threads=threads)
            t_mask += time.perf_counter() - t0
            n_rows += len(live)
            answers.append((np.asarray(live, dtype=np.int64), bits, k))
        elif op[0] == "ranges":
            t0 = time.perf_counter()
            answers.append(orc.get_range_batch(op[1], threads=threads))
            t_rng += time.perf_counter() - t0
            n_seq += len(op[1])
        elif op[0] == "locate":
            lo, hi, mx = op[1], op[2], op[3]
            rows = np.concatenate([np.arange(a, min(c, a + mx), dtype=np.uint64) for a, c in zip(lo, hi) if c > a] or [np.zeros(0, np.uint64)])
            rows = rows[::locate_stride]           # (a sample where the host's LF walks would take minutes: parity_check strides the GPU's answers alike)
            t0 = time.perf_counter()
            answers.append(orc.locate_bin_batch(rows, b, threads=threads))
            t_loc += time.perf_counter() - t0
            n_loc += len(rows)
        elif op[0] == "docs":
            d = op[1][::locate_stride]
            t0 = time.perf_counter()
            answers.append(orc.extract_batch_tokens(b[d], b[d + 1], threads=threads))    # get_doc = extract_text per document (index.py:68-75)
            t_doc += time.perf_counter() - t0
            n_doc += len(d)
        else:
            answers.append(None)
    return dict(mask_s=t_mask, ranges_s=t_rng, locate_s=t_loc, docs_s=t_doc, rows=n_rows, sequences=n_seq, located=n_loc, docs=n_doc), answers


def gpu_allowed_bits(index, ids, ff, kw, vocab=VOCAB):
    """the constraint the decode step applied, as the bitmap ``fmi_dev_allowed_bits`` returns for the same rows"""
    import ctypes
    from seal_amd._lib import check, lib
    rows, cur_len = ids.shape
    bits = torch.zeros(rows, (vocab + 31) // 32, dtype=torch.int32, device=ids.device)
    ff_arr = (ctypes.c_int64 * max(len(ff), 1))(*ff)
    check(lib().fmi_dev_allowed_bits(index.handle, torch.cuda.current_stream(ids.device).cuda_stream, rows, cur_len,
                                     ids.contiguous().data_ptr(), bits.data_ptr(), vocab, SHIFT, kw["pad"], kw["eos"], ff_arr, len(ff),
                                     kw["stop_at_count"], int(kw["always_allow_eos"])))
    torch.cuda.synchronize(ids.device)
    return bits.cpu().numpy().view(np.uint32)


def parity_check(index, trace, answers, vocab=VOCAB, locate_stride=1):
    """every recorded FM-index operation of one batch at BASELINE scale: what the GPU answered vs what the CPU oracle
    answers for the same operation on the same index -- bit-exact (ranges and counts of every key, the allowed-token
    set of every decode row = its distinct symbols, located positions + doc ids, extracted documents)"""
    ops = vals = bad = 0
    detail = {}

    def tally(name, n, nbad):
        nonlocal ops, vals, bad
        ops += 1
        vals += int(n)
        bad += int(nbad)
        d = detail.setdefault(name, {"ops": 0, "values": 0, "mismatches": 0})
        d["ops"] += 1; d["values"] += int(n); d["mismatches"] += int(nbad)
    for op, ans in zip(trace, answers):
        if op[0] == "mask":
            ids, ff, kw = op[1], op[2], op[3]
            assert kw["stop_at_count"] == 0 and not kw["always_allow_eos"]
            if len(op) > 4 and op[4] is not None:
                # the bitmap the decode step ACTUALLY applied (fmi_dev_last_constraint_bits: table / chained / generic form alike)
                got = op[4].cpu().numpy().view(np.uint32)
                applied = detail.setdefault("allowed_token_sets", {"ops": 0, "values": 0, "mismatches": 0})
                applied["source"] = "the bitmaps the timed path's constraint calls filled (captured per step), not a recomputation"
            else:
                got = gpu_allowed_bits(index, ids, ff, kw, vocab)
            live, bits, k = ans
            want = np.zeros_like(got)
            want[:, kw["pad"] >> 5] |= np.uint32(1 << (kw["pad"] & 31))       # finished rows: only pad (beam_search.py:119-127)
            want[live] = bits
            bad_rows = np.nonzero((got != want).any(axis=1))[0]
            tally("allowed_token_sets", got.shape[0], len(bad_rows))
            for r in bad_rows[:4]:          # what differs, for the log (a mismatch fails the run)
                x = np.unpackbits((got[r] ^ want[r]).view(np.uint8), bitorder="little")
                toks = np.nonzero(x)[0][:8].tolist()
                detail.setdefault("allowed_token_set_mismatch_samples", []).append(
                    {"cur_len": int(ids.shape[1]), "row": int(r), "ids": ids[r].tolist(), "forced": list(ff), "n_got": int(np.unpackbits(got[r].view(np.uint8)).sum()),
                     "n_want": int(np.unpackbits(want[r].view(np.uint8)).sum()), "differing_tokens": toks,
                     "in_got": [bool((got[r][t >> 5] >> (t & 31)) & 1) for t in toks]})
            pop = np.unpackbits(got[live].view(np.uint8), axis=1).sum(axis=1)
            tally("distinct_symbol_counts", len(live), int((pop != k.astype(np.int64)).sum()))
        elif op[0] == "ranges":
            lo, hi = ans
            tally("ranges_and_counts", 2 * len(lo), int((lo != op[2]).sum() + (hi != op[3]).sum()))
        elif op[0] == "locate":
            pos, doc = ans
            g_pos, g_doc = op[4][::locate_stride], op[5][::locate_stride]
            tally("located_positions_and_doc_ids", 2 * len(pos), int((pos.astype(np.int64) != g_pos).sum() + (doc.astype(np.int64) != g_doc).sum())
                  if len(pos) == len(g_pos) else max(len(pos), len(g_pos)))
        elif op[0] == "docs":
            flat, offs = ans
            g_flat, g_offs = op[2], op[3]
            if locate_stride > 1:              # the GPU's documents of the same sample: re-packed from its flat array
                o = np.asarray(op[3], dtype=np.int64)
                pick = np.arange(0, len(o) - 1, locate_stride)
                g_flat = np.concatenate([op[2][o[i]:o[i + 1]] for i in pick] or [np.zeros(0, dtype=np.asarray(op[2]).dtype)])
                g_offs = np.concatenate([[0], np.cumsum(o[pick + 1] - o[pick])])
            same_shape = len(flat) == len(g_flat) and np.array_equal(np.asarray(offs, dtype=np.int64), np.asarray(g_offs, dtype=np.int64))
            tally("extracted_document_tokens", len(flat), int((flat != g_flat).sum()) if same_shape else max(len(flat), len(g_flat)))
    return {"ops": ops, "values_compared": vals, "mismatches": bad, "by_kind": detail,
            "against": "oracle/fm_oracle.c (sdsl-style wt_int + rank_support_v, SA/32, ISA/64) built from this index's BWT"}


def _suffix_positions(index, rows):
    """SA[rows] as int64, straight from the index's resident suffix array (32 low bits + 8 high bits above 2^32 rows)"""
    sa_lo = device_array(index, "sa_lo", "<i4")
    sa_hi = device_array(index, "sa_hi", "|u1")
    pos = sa_lo[rows].long() & 0xFFFFFFFF
    if sa_hi is not None:
        pos |= sa_hi[rows].long() << 32
    return pos


def sa_audit(index, n_pairs=1 << 20, seed=7):
    """An audit of the GPU-BUILT index that does not go through anything the builder produced except the arrays under test
    (the NQ-scale oracle of ``build_cpu_oracle`` is fed this index's own BWT and SA samples, so it cannot see a mis-sorted
    suffix array): (1) ``n_pairs`` random adjacent row pairs: text[SA[i]:] < text[SA[i+1]:] by direct symbol-by-symbol
    comparison on the device; (2) sum(SA) == n(n-1)/2 (with (1): a permutation); (3) for ``n_pairs`` random rows i: one
    backward-search step with symbol text[SA[i]-1] from [i, i] must land on the single row j with SA[j] == SA[i]-1, i.e.
    BWT[i] == text[SA[i]-1] as the wavelet matrix sees it and LF(i) per C[] + rank are consistent with the suffix array."""
    import ctypes
    from seal_amd._lib import check, lib
    n = index.size()
    text = device_array(index, "text", "<i2" if index_sym_bytes(index) == 2 else "<i4")
    mask = 0xFFFF if text.dtype == torch.int16 else 0x7FFFFFFF
    dev = text.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    t0 = time.perf_counter()
    rows = torch.randint(0, n - 1, (n_pairs,), generator=g, device=dev)
    pa, pb = _suffix_positions(index, rows), _suffix_positions(index, rows + 1)
    ok = torch.zeros(n_pairs, dtype=torch.bool, device=dev)
    idx = torch.arange(n_pairs, device=dev)
    d, max_lcp = 0, 0
    while idx.numel():
        inside = (pa + d < n) & (pb + d < n)               # two distinct suffixes differ before either ends (unique sentinel)
        idx, pa, pb = idx[inside], pa[inside], pb[inside]
        a, b = text[pa + d].long() & mask, text[pb + d].long() & mask
        diff = a != b
        ok[idx[diff]] = a[diff] < b[diff]
        keep = ~diff
        idx, pa, pb = idx[keep], pa[keep], pb[keep]
        d += 1
        if idx.numel():
            max_lcp = d
    order_bad = int((~ok).sum())
    # (2) permutation checksum
    sa_lo = device_array(index, "sa_lo", "<i4")
    sa_hi = device_array(index, "sa_hi", "|u1")
    tot = 0
    CH = 1 << 28
    for a0 in range(0, n, CH):
        tot += int((sa_lo[a0:a0 + CH].long() & 0xFFFFFFFF).sum())
        if sa_hi is not None:
            tot += int(sa_hi[a0:a0 + CH].long().sum()) << 32
    sum_ok = tot == n * (n - 1) // 2
    # (3) BWT / LF consistency through the product's own backward-search step
    rows = torch.randint(0, n, (n_pairs,), generator=g, device=dev)
    p = _suffix_positions(index, rows)
    prev = torch.where(p == 0, torch.full_like(p, n - 1), p - 1)
    sym = (text[prev].long() & mask).contiguous()
    lo_out, hi_out = torch.empty_like(rows), torch.empty_like(rows)
    check(lib().fmi_dev_bs_step(index.handle, torch.cuda.current_stream(dev).cuda_stream, n_pairs, sym.data_ptr(), rows.data_ptr(), rows.data_ptr(),
                                lo_out.data_ptr(), hi_out.data_ptr()))
    torch.cuda.synchronize()
    single = lo_out == hi_out
    lf_bad = int((~single).sum()) + int((_suffix_positions(index, lo_out[single]) != prev[single]).sum())
    return {"adjacent_suffix_pairs": n_pairs, "suffix_order_violations": order_bad, "longest_common_prefix_seen": max_lcp,
            "sum_of_sa_is_n_choose_2": bool(sum_ok), "lf_rows": n_pairs, "bwt_lf_violations": lf_bad,
            "mismatches": order_bad + lf_bad + (0 if sum_ok else 1), "seconds": round(time.perf_counter() - t0, 2),
            "against": "direct comparison of text[SA[i]:] and text[SA[i+1]:] on the device; SA[LF(i)] == SA[i]-1 through fmi_dev_bs_step"}


def retrieval_title_length():
    from seal_amd import retrieval
    return retrieval.TITLE_MAX_LENGTH


def score_parity(searcher, model, index, queries, bias, dev, n_rescore_queries=4, tol=1e-4, fp32_model=None):
    """The float half of parity at the bench's own geometry (BART-large, beam 15, batch 20): the body and title decodes of one
    batch are run once more exactly as the searcher issues them (retrieval.py:70-83,162-176), every hypothesis score the beam
    loop recorded is recomputed through HF's own cache-free fp32 forward (oracle/hf_scores.py), and the prefix-tree
    rescoring of a sample of the body keys is held to one HF row per key (reference keys.py:64-141)."""
    from oracle.hf_scores import compare_beam_history, compare_rescoring
    from seal_amd.beam_search import fm_index_generate, fm_index_generate_joint
    from seal_amd.keys import _pad_batch
    s = searcher
    cfg = model.config
    mk = s.marker_token_ids
    out = {}
    B = len(queries)
    kinds = (("beam_scores_body", "body", dict(max_length=s.length)),
             ("beam_scores_title", "title", dict(max_length=retrieval_title_length(), force_decoding_from=[s.title_bos_token_id],
                                                 eos_token_id=s.title_eos_token_id)))
    toks = {kind: [q[:-1] + mk[kind] + mk["+"] + q[-1:] for q in queries] for _, kind, _ in kinds}
    if s.joint_decode:
        # what the searcher runs: both decodes as ONE loop of 2 x batch x beams rows (fm_index_generate_joint)
        enc_ids = _pad_batch(toks["body"] + toks["title"], cfg.pad_token_id, dev)
        enc_mask = (enc_ids != cfg.pad_token_id).long()
        pend = fm_index_generate_joint(model, index, enc_ids, enc_mask, [dict(batch=B, **kw) for _, _, kw in kinds], num_beams=s.beam,
                                       length_penalty=s.length_penalty, logit_bias=torch.cat([bias, bias]))
        runs = [(name, pg, enc_ids[i * B:(i + 1) * B], enc_mask[i * B:(i + 1) * B]) for i, ((name, _, _), pg) in enumerate(zip(kinds, pend))]
    else:
        runs = []
        for name, kind, kw in kinds:
            enc_ids = _pad_batch(toks[kind], cfg.pad_token_id, dev)
            enc_mask = (enc_ids != cfg.pad_token_id).long()
            runs.append((name, fm_index_generate(model, index, enc_ids, enc_mask, min_length=1, length_penalty=s.length_penalty, num_beams=s.beam,
                                                 keep_history=True, logit_bias=bias, pending=True, **kw), enc_ids, enc_mask))
    assert model._seal_step_decoder._st.fused, "the fused step decoder must be the one that ran"
    for name, pg, enc_ids, enc_mask in runs:
        steps, final, nb, K, _ = pg._args
        out[name] = compare_beam_history(model, enc_ids, enc_mask, steps, final, nb, K, logit_bias=bias, tol=tol)
        out[name]["decoded_as"] = "one loop with the other decode (joint)" if s.joint_decode else "its own loop"
        if fp32_model is not None:
            # a reduced-precision decode (configs[4]) against the FP32 answer as well: distance, HF's own low-precision distance, rank changes
            from oracle.hf_scores import compare_with_fp32_forward
            out[name]["vs_fp32"] = compare_with_fp32_forward(model, fp32_model, enc_ids, enc_mask, steps, final, nb, K, logit_bias=bias)
    body_hyps = runs[0][1].result()
    nq = min(n_rescore_queries, len(queries))
    strip_ids = s.strip_token_ids
    keys = []
    for h in body_hyps[:nq]:
        kk = [(sc, k[1:] if k[0] in strip_ids else k) for sc, k in h if k]
        keys.append([(sc, k) for sc, k in kk if k])
    out["rescore_scores"] = compare_rescoring(model, [q for q in queries[:nq]], keys, tol=tol, length_penalty=0.0, logit_bias=bias[:nq],
                                              strip_from_bos=[s.title_bos_token_id, s.code_bos_token_id, cfg.decoder_start_token_id],
                                              strip_from_eos=[s.title_eos_token_id, s.code_eos_token_id, cfg.eos_token_id])
    return out


def kernel_source_sha256(root=ROOT) -> str:
    """what a PMC file must have been taken over to be cited (tools/summarize_pmc.py records the same digest)"""
    import hashlib
    csrc = os.path.join(root, "seal_amd", "csrc")
    return hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in ("fmi_kernels.hip", "fmi_device.h", "fmi_internal.h"))).hexdigest()


def cite_traffic(workload_tag, root=ROOT, now=None):
    """(HBM bytes per constraint call, where the figure comes from) from the newest profiles/r*_pmc_fetch_size*.json taken over the CURRENT
    kernel sources on THIS workload; (None, why not) otherwise -- counters of another kernel generation or workload are refused"""
    import glob
    traffic = traffic_src = None
    try:
        now = now or kernel_source_sha256(root)
        files = sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_fetch_size*.json")), reverse=True)
        for f in files:
            pmc = json.load(open(f))
            if pmc.get("_kernel_source_sha256") != now:
                continue            # counters of another kernel generation are REFUSED (the file records the source it profiled)
            if pmc.get("_workload") != workload_tag:
                continue            # ... and so are counters taken on another workload (index size, beam): per-workload files
            # every launch of the constraint calls (k_constrain_rows + k_constrain of a row-first call; k_constrain_table + k_table_bits of a
            # decode's first step), summed over the run and divided by the CALLS (one k_constrain or one k_constrain_table each)
            # (+ k_beam_advance: the rows' chains and list steps of the chained calls ride in that launch; its bookkeeping traffic comes along)
            mine = {k: v["FETCH_SIZE"] for k, v in pmc.items() if isinstance(v, dict) and "FETCH_SIZE" in v and ("k_constrain" in k or "k_table_bits" in k or "k_beam_advance" in k)}
            calls = sum(c["launches"] for k, c in mine.items() if "k_constrain<" in k or "k_constrain_table" in k)
            kib = sum(c["sum"] for c in mine.values()) / calls if calls else 0
            if kib:
                traffic = round(kib * 1024.0 * 2, 1)
                traffic_src = {"file": os.path.relpath(f, root), "builder_run": True, "commit": pmc.get("_commit"), "kernel_source_sha256": now[:16], "workload": workload_tag}
                break
        if traffic is None:
            traffic_src = {"refused": "no profiles/r*_pmc_fetch_size*.json was taken over the current fmi_kernels.hip / fmi_device.h / fmi_internal.h "
                                      "(sha256 %s) on this workload (%s); run tools/prof_bench.sh" % (now[:16], workload_tag), "candidates": [os.path.relpath(f, root) for f in files[:3]]}
    except Exception as e:
        traffic_src = {"error": repr(e)}
    return traffic, traffic_src


CALL_FORMS = {0: "generic", 1: "row_first", 2: "table", 3: "chained", 4: "advance", 5: "advance+chains"}


def read_call_log(handle, cap=4096):
    """the library's per-call log (fmi_dev_call_log): [{cur_len, rows, form, us, blocks}] in launch order; clears it"""
    import ctypes as C
    from seal_amd._lib import check, lib
    cl, rw, kd = (C.c_uint32 * cap)(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
    us, bl, n = (C.c_float * cap)(), (C.c_uint64 * cap)(), C.c_uint64()
    check(lib().fmi_dev_read_call_log(handle, cap, cl, rw, kd, us, bl, C.byref(n)))
    return [{"cur_len": cl[i], "rows": rw[i], "form": CALL_FORMS.get(kd[i], str(kd[i])), "us": float(us[i]), "blocks": int(bl[i])} for i in range(min(cap, n.value))]


def merge_call_logs(timed, counted, fused=None):
    """One record per CONSTRAINT call of one un-overlapped batch, in launch order.  Three passes over the same batch supply it:
    `timed` -- events around every launch, k_beam_advance run as two launches (its bookkeeping, then the rows' chains of the next call:
    "advance" / "advance+chains" records, logged under the cur_len of the call they precede); `counted` -- the same launches with the
    in-kernel block counters, drained call by call; `fused` -- the product's launch shape, k_beam_advance ONE launch, events around it.
    A call's `us` = its own launches + `chains_us`, what its chains ADD to the k_beam_advance launch they ride in (fused duration minus
    the bookkeeping launch's duration; never below zero); `chains_alone_us` = the chains as a launch of their own, an upper bound that
    pays for a launch the product does not make.  `MB` = the call's blocks + the chains'.  [] when the passes disagree about what ran."""
    def shape(log):
        return [(a["cur_len"], a["rows"], a["form"]) for a in log]
    if not timed or shape(timed) != shape(counted):
        return [], []
    fused_adv = {}
    if fused:
        for a in fused:
            if a["form"] in ("advance", "advance+chains"):
                fused_adv[a["cur_len"]] = a["us"]
    calls, other = [], []
    pending, book = None, {}
    for a, b in zip(timed, counted):
        rec = {"cur_len": a["cur_len"], "rows": a["rows"], "form": a["form"], "MB": b["blocks"] * 128.0 / 1e6, "us": a["us"]}
        if a["form"] == "advance+chains":
            pending = rec
            continue
        if a["form"] == "advance":
            book[a["cur_len"]] = a["us"]
            other.append({"cur_len": a["cur_len"], "rows": a["rows"], "form": a["form"], "us": round(a["us"], 2),
                          **({"fused_with_chains_us": round(fused_adv[a["cur_len"]], 2)} if a["cur_len"] in fused_adv else {})})
            continue
        if pending is not None and pending["cur_len"] == rec["cur_len"]:
            alone = pending["us"]
            cl = rec["cur_len"]
            added = max(0.0, fused_adv[cl] - book[cl]) if (cl in fused_adv and cl in book) else alone
            rec["own_launches_us"] = round(rec["us"], 2)
            rec["chains_us"] = round(added, 2)
            rec["chains_alone_us"] = round(alone, 2)
            rec["chains_MB"] = round(pending["MB"], 3)
            rec["us"] += added
            rec["us_upper_bound"] = round(rec["own_launches_us"] + alone, 2)
            rec["MB"] += pending["MB"]
        pending = None
        calls.append(rec)
    for c in calls:
        us, mb = c["us"], c["MB"]
        c["MB"], c["us"] = round(mb, 3), round(us, 2)
        c["GBps"] = round(mb * 1e6 / (us * 1e-6) / 1e9, 1) if us > 0 else None
        c["frac"] = round(mb * 1e6 / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if us > 0 else None
    return calls, other


AGG_STAGES = ["k_agg_locate", "sort_by_position(rocprim)", "documents+coverage(k_occ_prepare+k_mis)", "(sort_by_document: gone)",
              "entry_boundaries(k_heads+scan+k_entry_starts)", "k_entries", "ranking(k_sel_minmax+k_sel_hist x2+k_sel_compact+k_sel_final)", "token_tables(memsets+scatters)",
              "k_full_score(ranked documents)", "k_rank_docs", "k_full_score(top-k records)"]


def read_agg_timing(handle):
    import ctypes as C
    from seal_amd._lib import check, lib
    ms, cnt, calls = (C.c_double * len(AGG_STAGES))(), (C.c_uint64 * 5)(), C.c_uint64()
    check(lib().fmi_dev_read_agg_timing(handle, ms, cnt, C.byref(calls)))
    return {"stage_ms": list(ms), "counts": list(cnt), "calls": calls.value}


def aggregate_roofline(t, index, workload_tag=None):
    """`roofline_aggregate`: the locate + doc-binning + evidence kernels of ONE un-overlapped batch (north_star: "then locate() + doc-id
    binning for scoring"; reference keys.py:314-350, index.py:77-82), HIP events after every stage of fmi_dev_aggregate.  Algorithmic bytes
    = the arrays a stage must read and write once for the rows / entries / documents it processed (DESIGN.md 5.4 derives each figure)."""
    if not t or not t["calls"]:
        return None
    rows, entries, docs, kept, doc_tok = t["counts"]
    wide_sa = index.size() > (1 << 32)
    per = {
        # row -> SA[row] (4 B, +1 B above 2^32 rows); writes the sort key 8 + the occurrence number 4
        "k_agg_locate": rows * ((5 if wide_sa else 4) + 12),
        # k_occ_prepare, in position order: key 8 + occurrence number 4 -> rare key 4 + document 4 (sampled table 4 + boundary 8, shared by
        # neighbours: counted once per row all the same) + window 2 + state 1; k_mis: E 8 + M 2 + PRI 4 + state 1 -> newflag 1
        "documents+coverage(k_occ_prepare+k_mis)": rows * (8 + 4 + 4 + 8 + 4 + 4 + 2 + 1 + 8 + 2 + 4 + 1 + 1),
        # per located row: occurrence number 4 + rare key 4 + newflag 1, read once, written once in processing order, read again (x 3);
        # per (query, document) entry: key 8 + document 4 + nkeys 4 + rank 8 + first 4 + q 4 + doc 4 + score 8 + best 4 + one key (4 + 8)
        "k_entries": rows * 27 + entries * 60,
        # the document's tokens from the resident text (2 B each) + its score (8 B); everything else lives in LDS
        "k_full_score(ranked documents)": doc_tok * 2 + docs * 8,
    }
    out = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "calls": t["calls"], "located_rows": rows, "document_entries": entries,
           "documents_scored": docs, "documents_kept": kept, "document_tokens": doc_tok, "stages": []}
    for name, ms in zip(AGG_STAGES, t["stage_ms"]):
        rec = {"stage": name, "us": round(ms * 1e3, 1)}
        if name in per and ms > 0:
            gbps = per[name] / (ms * 1e-3) / 1e9
            rec.update({"algorithmic_MB": round(per[name] / 1e6, 2), "achieved": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4)})
        out["stages"].append(rec)
    loc = out["stages"][0]
    out.update({"kernel": "k_agg_locate", "achieved": loc.get("achieved"), "frac": loc.get("frac"), "total_us": round(sum(t["stage_ms"]) * 1e3, 1),
                "measured_on": "the un-overlapped batch after the timed region that also times the constraint calls",
                "note": "k_full_score is an LDS / ALU kernel (hash-trie matching, rank sort and greedy cover in LDS): its HBM bytes are the documents' tokens; "
                        "the radix sorts are rocPRIM's"})
    out["traffic"], out["traffic_source"] = cite_traffic_agg(index, workload_tag=workload_tag)
    loc_pmc = (out["traffic"] or {}).get("k_agg_locate") if isinstance(out["traffic"], dict) else None
    if loc_pmc and loc.get("algorithmic_MB"):
        # k_agg_locate's memory-side bytes per batch over its algorithmic bytes: FETCH_SIZE x 2 (the guide's gfx950 correction) + WRITE_SIZE;
        # `_uncorrected`: FETCH_SIZE as reported (its small gathers are not the access width the x 2 was calibrated on)
        out["traffic_ratio"] = round((2 * loc_pmc["fetch_MB"] + loc_pmc["write_MB"]) / loc["algorithmic_MB"], 2)
        out["traffic_ratio_uncorrected"] = round((loc_pmc["fetch_MB"] + loc_pmc["write_MB"]) / loc["algorithmic_MB"], 2)
    # the document look-up moved out of k_agg_locate in round 6 (into k_occ_prepare, behind the sort): its two random sectors per row with it --
    # said here so that the ratio above is not read as "the traffic is gone"
    occ_pmc = (out["traffic"] or {}).get("k_occ_prepare") if isinstance(out["traffic"], dict) else None
    if occ_pmc:
        occ_alg = rows * (8 + 4 + 4 + 8 + 4 + 4 + 2 + 1) / 1e6
        out["doc_binning_kernel"] = "k_occ_prepare"
        out["doc_binning_traffic_ratio"] = round((2 * occ_pmc["fetch_MB"] + occ_pmc["write_MB"]) / occ_alg, 2)
        out["doc_binning_traffic_ratio_uncorrected"] = round((occ_pmc["fetch_MB"] + occ_pmc["write_MB"]) / occ_alg, 2)
    return out


def cite_traffic_agg(index=None, root=ROOT, workload_tag=None):
    """FETCH_SIZE + WRITE_SIZE of the aggregation kernels from the newest builder-run profiles/r*_pmc_agg*.json taken over the current
    fmi_aggregate.hip (same refusal rule as cite_traffic)"""
    import glob, hashlib
    try:
        sha = hashlib.sha256(open(os.path.join(root, "seal_amd", "csrc", "fmi_aggregate.hip"), "rb").read()).hexdigest()
        for f in sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_agg*.json")), reverse=True):
            pmc = json.load(open(f))
            if pmc.get("_aggregate_source_sha256") == sha and (workload_tag is None or pmc.get("_workload") == workload_tag):
                per = dict(pmc.get("per_batch_MB") or {})
                for name, rec in (pmc.get("per_kernel_per_batch") or {}).items():
                    if name.startswith("k_agg_locate") or name.startswith("k_occ_prepare"):
                        per[name.split("(")[0]] = {"fetch_MB": rec["fetch_MB"], "write_MB": rec["write_MB"]}
                return per, {"file": os.path.relpath(f, root), "builder_run": True, "aggregate_source_sha256": sha[:16]}
        return None, {"refused": "no profiles/r*_pmc_agg*.json taken over the current fmi_aggregate.hip (sha256 %s) on this workload (%s)" % (sha[:16], workload_tag)}
    except Exception as e:
        return None, {"error": repr(e)}


def _prefix_table_stats(index):
    """the per-token node tables the index handle built for the first constrained step of the decodes"""
    import ctypes
    from seal_amd._lib import check, lib
    t, n, b = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    check(lib().fmi_dev_prefix_table_stats(index.handle, ctypes.byref(t), ctypes.byref(n), ctypes.byref(b)))
    return {"tables": t.value, "leaf_level_nodes": n.value, "hbm_mib": round(b.value / 2**20, 1)}


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n_gpus: int, argv, port: int = None, script: str = None):
    """the command line `python bench.py --gpus N ...` turns itself into when no launcher set RANK / WORLD_SIZE: one rank per GPU of
    this node under torch.distributed.run (rendezvous on 127.0.0.1: the container's hostname may not resolve)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), script or os.path.abspath(__file__)] + list(argv)


def launch_ranks(n_gpus: int, argv) -> int:
    """start the N ranks and wait for them; rank 0's JSON line goes to this process's stdout, every rank's log to stderr"""
    import subprocess
    cmd = launch_command(n_gpus, argv)
    print("[bench] --gpus %d without RANK/WORLD_SIZE in the environment: launching %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL over dmabuf IPC (the host driver supports nothing else)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_gpus)))
    return subprocess.call(cmd, env=env)


def dry_run_launch(args) -> int:
    """--dry-run-launch: every rank joins the process group (nccl = RCCL when it has a GPU, gloo otherwise), all-reduces a one, and rank 0
    prints the launch fields of the line.  No index, no model: what it proves is that `--gpus N` yields N ranks and one line."""
    import torch.distributed as dist
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if gpu else torch.device("cpu")
    backend = "nccl" if gpu else "gloo"
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if gpu else {}))
    t = torch.ones(1, device=dev, dtype=torch.float64)
    dist.all_reduce(t)
    print(f"[bench] rank {rank} of {dist.get_world_size()} up (backend {dist.get_backend()}, device {dev})", file=sys.stderr, flush=True)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launch dry run", "n_gpus": dist.get_world_size(), "ranks_reduced": int(t.item()), "requested_gpus": args.gpus,
                          "config": {"parallelism": f"query-sharded x{dist.get_world_size()}, backend {dist.get_backend()}"}, "dry_run": True}), flush=True)
    dist.destroy_process_group()
    return 0 if int(t.item()) == world else 4


def cpu_smoke(args) -> int:
    """--cpu-smoke: the N > 1 plumbing of this file end to end WITHOUT a GPU -- launcher (above) -> one rank per "GPU" -> process group
    (gloo) -> this rank's contiguous block of the queries (seal_amd.distributed.shard_queries) -> the product's SEALSearcher.batch_search
    -> pack_topk -> ONE all_gather (gather_topk) -> barrier, max-over-ranks time, ONE line on rank 0 with the same launch fields as the
    real line (n_gpus, ranks_seen, per-rank figures) and the gathered top-k as hex, so that a test can hold N = 2 to N = 1 bit for bit.
    A TEST MODE, not a measurement: the searcher is the tiny CPU one of tests/test_distributed_gloo.py whose index queries are answered
    by the oracle (tests/ infrastructure); nothing here is timed for the record and the line says so."""
    import torch.distributed as dist
    from seal_amd.distributed import gather_topk, pack_topk, shard_bounds, shard_queries
    from tests.test_distributed_gloo import _cpu_searcher
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.set_num_threads(2)
    use_dist = world > 1 or bool(os.environ.get("SEAL_BENCH_FORCE_DIST"))
    if use_dist:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    searcher, queries = _cpu_searcher(batch_size=1)
    mine = shard_queries(queries, rank, world)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    res = searcher.batch_search(mine, k=args.topk, detokenize=False) if mine else []
    top = gather_topk(pack_topk(res, args.topk), len(queries))
    if use_dist:
        dist.barrier()
    mine_s = time.perf_counter() - t0
    t = torch.tensor([mine_s], dtype=torch.float64)
    per_rank = [torch.tensor([len(mine) / mine_s], dtype=torch.float64)]
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([len(mine) / mine_s], dtype=torch.float64))
    if rank == 0:
        import hashlib
        raw = top.contiguous().numpy().tobytes()
        print(json.dumps({"metric": "cpu smoke of the N-rank path (launcher, shard, search, top-k gather, one line): a test mode, NOT a measurement",
                          "cpu_smoke": True, "value": round(len(queries) / float(t.item()), 3), "unit": "queries/s", "n_gpus": world,
                          "ranks_seen": dist.get_world_size() if use_dist else 1, "requested_gpus": args.gpus, "queries": len(queries),
                          "shards": [list(shard_bounds(len(queries), r, world)) for r in range(world)],
                          "per_rank_queries_per_s_min": round(min(float(x) for x in per_rank), 3),
                          "per_rank_queries_per_s_max": round(max(float(x) for x in per_rank), 3),
                          "topk_shape": list(top.shape), "topk_sha256": hashlib.sha256(raw).hexdigest(), "topk_hex": raw.hex(),
                          "config": {"workload": "tests/golden/ref_searcher.json corpus, tiny BART on CPU, index queries answered by the oracle",
                                     "parallelism": f"query-sharded x{world}, backend {'gloo' if use_dist else 'none'}"}}), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=int(os.environ.get("SEAL_BENCH_DOCS", 21015324)))
    ap.add_argument("--batch", type=int, default=20)
    ap.add_argument("--beam", type=int, default=15)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline leg (0 = all the cores this process may run on, SURVEY.md 8(d))")
    ap.add_argument("--jobs", type=int, default=1, help="host worker processes (only used by the host aggregation routines)")
    ap.add_argument("--no-overlap", action="store_true", help="do not enqueue the next batch's decodes ahead of this batch's rescoring/aggregation")
    ap.add_argument("--no-joint-decode", action="store_true", help="body and title decodes as two loops of batch x beams rows (the reference's "
                    "order) instead of one loop of 2 x batch x beams rows")
    ap.add_argument("--workload", choices=["nq", "stress"], default="nq", help="nq: BASELINE configs[1] (configs[3] with --docs 36000000), the "
                    "contract's default; stress: configs[4], the same complete search over 100 M synthetic passages (1.4e10 symbols: 40-bit suffix "
                    "array sorted in slices, fmi_build_device_sliced) with beam 30 and a bf16 BART-large")
    ap.add_argument("--slice-rows", type=int, default=1 << 29, help="stress workload: suffixes per slice of the suffix-array construction")
    ap.add_argument("--bf16-tol", type=float, default=0.25, help="stress workload: tolerance of the bf16 hypothesis scores against HF's bf16 forward")
    ap.add_argument("--cpu-locate-sample", type=int, default=0, help="CPU oracle replay: every k-th located row / every k-th document only "
                    "(0: all at nq, 64 at stress: sdsl's sampled suffix array costs ~31 LF steps per row on the host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--keys-oracle-queries", type=int, default=2, help="queries of the recorded batch whose aggregation is also held to "
                    "oracle/keys_oracle.py (python loops over every located row: seconds per query)")
    ap.add_argument("--latency-batches", type=int, default=20, help="un-pipelined single batches timed AFTER the timed region for the p50 latency")
    ap.add_argument("--with-other-depth", action="store_true", help="also time the same batches through the other retrieval depth "
                    "(first stage only <-> complete search); first-stage-only aggregates on the host and is slow on the phrase corpus")
    ap.add_argument("--corpus-phrases", type=int, default=int(os.environ.get("SEAL_BENCH_PHRASES", 20000000)),
                    help="documents assembled from a dictionary of P repeating phrases (default 20 M: n-grams repeat as in real text, a key "
                         "locates ~3e5 rows per query like a real NQ index); 0 = the i.i.d. Zipf corpus of round 1 (~1e4 rows per query)")
    ap.add_argument("--no-query-keys", action="store_true", help="leave out the query n-gram keys (add_query_to_keys, the reference's default)")
    ap.add_argument("--first-stage-only", action="store_true",
                    help="stop after the first retrieval stage (SURVEY.md 8d metric) instead of the reference's complete batch_search")
    ap.add_argument("--dry-run-launch", action="store_true", help="start the ranks, form the process group (nccl = RCCL with a GPU, gloo without), "
                    "all-reduce a one per rank and print the line's launch fields only: checks the N-rank launch without building an index")
    ap.add_argument("--cpu-smoke", action="store_true", help="TEST MODE, no GPU: the launcher -> ranks -> shard -> search -> top-k gather -> one "
                    "line path over the tiny CPU searcher of tests/test_distributed_gloo.py (gloo); the line carries the gathered top-k as hex")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver types it: nobody started the ranks yet -> this process becomes the launcher
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run_launch:
        sys.exit(dry_run_launch(args))
    if args.cpu_smoke:
        sys.exit(cpu_smoke(args))
    # the contract is ONE JSON line on stdout: RCCL prints a version banner there (seen with one rank: five lines after
    # the JSON line), so file descriptor 1 is pointed at stderr for everything but the line itself
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] rank {rank}: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is what runs", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        print(f"[bench] bench.py needs a GPU (rank {rank} of {world})", file=sys.stderr, flush=True)
        if world > 1:
            # the launcher tears the other ranks down as soon as one fails: linger so that every rank's own diagnosis reaches the log
            time.sleep(float(os.environ.get("SEAL_BENCH_FAIL_LINGER_S", "2")))
        sys.exit(5)
    # host budget of a rank: the search step is one python thread + the key-scoring threads of fmi_agg_score_pack; all
    # ranks of a node share its cores (the CPU baseline leg alone uses --cpu-threads, on rank 0 at N=1 only)
    cores = os.cpu_count() or 1
    host_threads = max(1, min(8, cores // (2 * max(world, 1))))
    os.environ.setdefault("SEAL_HOST_THREADS", str(host_threads))
    torch.set_num_threads(max(1, min(8, cores // max(world, 1))))
    try:
        log("host memory: " + next(l for l in open("/proc/meminfo") if l.startswith("MemTotal")).strip())
    except Exception:
        pass
    log(f"host: {cores} cores / {world} rank(s): {os.environ['SEAL_HOST_THREADS']} key-scoring threads, {torch.get_num_threads()} torch CPU threads per rank")
    if world > 1 and hasattr(os, "sched_setaffinity"):
        # one rank per GPU on a shared host: every rank keeps to its own contiguous share of the cores (python thread,
        # key-scoring threads, torch CPU threads), so that 8 ranks do not migrate over each other's caches / NUMA nodes
        avail = sorted(os.sched_getaffinity(0))
        per = max(1, len(avail) // world)
        mine = avail[(local % world) * per:(local % world + 1) * per] or avail
        try:
            os.sched_setaffinity(0, mine)
            log(f"rank 0 of {world}: pinned to cores {mine[0]}..{mine[-1]} ({len(mine)} of {len(avail)})")
        except OSError:
            pass
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # A run that stops making progress (a GPU-side stall: the host then waits in a stream synchronize for ever) ends itself
    # with the stacks of all threads instead of sitting on the GPU until somebody's outer limit: the default run takes ~90 s,
    # the stress tier ~6 min.  SEAL_BENCH_WATCHDOG_S=0 turns it off.
    import faulthandler
    stress = args.workload == "stress"
    limit = float(os.environ.get("SEAL_BENCH_WATCHDOG_S", 2400 if stress else 900 + 20 * max(0, args.steps - 20)))
    if limit > 0:
        faulthandler.dump_traceback_later(limit, exit=True)
    if stress:
        assert world == 1, "the stress workload is a single-GPU measurement"
        if "--beam" not in sys.argv:
            args.beam = 30
        if "--docs" not in sys.argv and "SEAL_BENCH_DOCS" not in os.environ:
            args.docs = 100_000_000
        if not args.cpu_locate_sample:
            args.cpu_locate_sample = 64
    force_dist = bool(os.environ.get("SEAL_BENCH_FORCE_DIST"))     # exercise the RCCL path with one rank
    use_dist = world > 1 or force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    import __graft_entry__ as ge
    from seal_amd import FMIndex
    from seal_amd._lib import check, lib
    from seal_amd.retrieval import SEALSearcher
    from seal_amd import retrieval, keys as rk
    from seal_amd.distributed import gather_topk, pack_topk
    if rank == 0:
        ge.build()
    if use_dist:
        dist.barrier()

    t0 = time.perf_counter()
    data, beg, title_len, ids_by_rank = synth_corpus(args.docs, dev, seed=0, phrases=args.corpus_phrases, text16=stress)
    torch.cuda.synchronize()
    log(f"corpus: {args.docs} docs, {data.numel() - (1 if stress else 0)} symbols in {time.perf_counter() - t0:.1f}s")
    workload_tag = f"{args.workload}-{args.docs}"          # a PMC traffic file is only cited by lines of the workload it was taken on
    log(f"workload_tag={workload_tag}")
    n_batches = args.warmup + args.steps + 1
    queries, bias = synth_queries(n_batches * args.batch, data, beg, title_len, ids_by_rank, dev, seed=1 + rank)
    t0 = time.perf_counter()
    index = FMIndex()
    if stress:
        # the corpus IS the resident text (nothing copied: no room for a second 28 GB); the suffix array is sorted in slices
        torch.cuda.empty_cache()
        index.initialize_from_device_text(data, beg.tolist(), slice_rows=args.slice_rows)
        _text = device_array(index, "text", "<i2")
        text_is_input = _text.data_ptr() == data.data_ptr() and bool(_text[-1] == 0)
    else:
        index.initialize_from_device(data, beg.tolist())
        # the index's resident text must be the corpus it was given (+ the sentinel): what sa_audit compares suffixes of
        _text = device_array(index, "text", "<i2" if index_sym_bytes(index) == 2 else "<i4")
        text_is_input = bool(_text[-1] == 0)
        for a0 in range(0, data.numel(), 1 << 28):
            b0 = min(data.numel(), a0 + (1 << 28))
            text_is_input &= bool(torch.equal(_text[a0:b0].to(torch.int32) & (0xFFFF if _text.dtype == torch.int16 else 0x7FFFFFFF), data[a0:b0]))
    index.labels = None
    del data, _text
    torch.cuda.empty_cache()
    index_build_s = time.perf_counter() - t0
    log(f"index: n={index.size()} levels={lib().fmi_levels(index.handle)} HBM={index.device_bytes() / 2**30:.1f} GiB "
        f"built on GPU in {index_build_s:.1f}s")

    from transformers import BartConfig, BartForConditionalGeneration
    t0 = time.perf_counter()
    torch.manual_seed(0)
    cfg = BartConfig()
    cfg.forced_bos_token_id = None
    with torch.device(dev):
        model = BartForConditionalGeneration(cfg)
    model.eval()
    with torch.no_grad():
        for tok in (cfg.pad_token_id, cfg.bos_token_id, VOCAB - 1):      # reference retrieval.py:584-588
            model.final_logits_bias[0, tok] = float("-inf")
    if stress:
        model.to(torch.bfloat16)          # BASELINE configs[4]: bf16 BART decode (fused kernels: bf16 storage, fp32 accumulation)
    log(f"BART-large random init ({'bf16' if stress else 'fp32'}) in {time.perf_counter() - t0:.1f}s")

    searcher = SEALSearcher(index, None, model, add_query_to_keys=not args.no_query_keys, detokenize=False, first_stage_only=args.first_stage_only,
                            beam=args.beam, batch_size=args.batch, jobs=args.jobs, overlap=not args.no_overlap,
                            joint_decode=not args.no_joint_decode)
    from seal_amd.bart_decoder import BartStepDecoder
    model._seal_step_decoder = BartStepDecoder(model)
    handles = [index.handle]
    # The warm-up and the TIMED region run the product: no in-kernel probe counters, no event pairs around the constraint calls
    # (round 3 timed the counting instantiation of k_constrain).  Launch times and block counts for the roofline come from two
    # separate un-overlapped passes over one more batch of the same workload, after the timed region (below).

    def read_counters(stats=None):
        """(probes, launches, kernel ms) summed over the handles since the last read; expand stats accumulate"""
        import ctypes as C
        tp, tl, tk = 0, 0, 0.0
        for hd in handles:
            if stats is not None:
                xs = (C.c_uint64 * 4)()
                check(lib().fmi_dev_read_expand_stats(hd, xs))
                for j in range(4):
                    stats[j] += xs[j]
            pr, ln, km = C.c_uint64(), C.c_uint64(), C.c_double()
            check(lib().fmi_dev_read_probe_count(hd, C.byref(pr)))
            check(lib().fmi_dev_read_timing(hd, C.byref(ln), C.byref(km)))
            tp += pr.value; tl += ln.value; tk += km.value
        return tp, tl, tk

    def run_batches(i0, n):
        """n consecutive batches in ONE searcher call: key generation of batch i+1 (GPU) overlaps the
        host-side evidence aggregation of batch i (worker threads), as in the reference's imap pipeline"""
        lo, hi = i0 * args.batch, (i0 + n) * args.batch
        searcher.logit_bias = bias[lo:hi]
        res = searcher.batch_search(queries[lo:hi], k=args.topk)
        # the path's only exchange: top-k (doc id, score) of every query to every rank (RCCL all_gather)
        top = pack_topk(res, args.topk)
        top = gather_topk(top.to(dev) if use_dist else top, world * n * args.batch, device=dev)
        return top, res

    def run_batch(i):
        return run_batches(i, 1)

    # the index keeps `beginnings` as a 21M-element python list (reference API); keep the cyclic GC
    # from re-scanning it (and the model) on every full collection
    import gc
    gc.collect()
    gc.freeze()
    # W warm-up steps: the first ones as single batches (their times are reported), the last ones -- when W >= 3 -- as ONE pipelined call, which is what
    # the timed region is: the overlapped loop's own first-use work (streams, fences, the decodes enqueued two batches ahead, the rescoring graph's buckets
    # on the rescoring stream) then falls into the warm-up, not into the first batches of the timed call (SEAL_BENCH_SINGLE_WARMUP=1: single batches only)
    step_ms = []                                          # un-pipelined single-batch latency
    n_single = args.warmup if (args.warmup < 3 or os.environ.get("SEAL_BENCH_SINGLE_WARMUP") == "1") else args.warmup - max(2, args.warmup // 2)
    for i in range(n_single):
        t1 = time.perf_counter()
        run_batch(i)
        torch.cuda.synchronize()
        step_ms.append((time.perf_counter() - t1) * 1e3)
    if n_single < args.warmup:
        run_batches(n_single, args.warmup - n_single)
        torch.cuda.synchronize()
    import ctypes
    # (the warm-up batches' results are garbage by now: collected and the survivors frozen, like the index's lists above, so that no full
    #  collection of theirs falls into the timed call)
    gc.collect()
    gc.freeze()

    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t_start = time.perf_counter()
    top, res = run_batches(args.warmup, args.steps)       # exactly K steps (batches), pipelined
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    my_elapsed = elapsed
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # p50 latency: `--latency-batches` single batches, one at a time (nothing enqueued ahead: what one caller of batch_search waits for),
    # AFTER warm-up and the timed region -- the timed region's own batches run once more, each bracketed by a device synchronise
    lat_ms = []
    for j in range(max(0, args.latency_batches)):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_batch(args.warmup + j % max(1, args.steps))
        torch.cuda.synchronize()
        lat_ms.append((time.perf_counter() - t1) * 1e3)
    lat = torch.tensor([float(np.median(lat_ms)) if lat_ms else 0.0, float(np.percentile(lat_ms, 90)) if lat_ms else 0.0], dtype=torch.float64, device=dev)
    my_qps = torch.tensor([args.batch * args.steps / my_elapsed, index_build_s], dtype=torch.float64, device=dev)
    per_rank = [my_qps.tolist()]
    if use_dist:
        dist.all_reduce(lat, op=dist.ReduceOp.MAX)
        gathered = [torch.zeros_like(my_qps) for _ in range(world)]
        dist.all_gather(gathered, my_qps)
        per_rank = [g.tolist() for g in gathered]
    import resource
    peak_rss_gib = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20        # this rank's peak host RSS so far (KiB -> GiB)
    if use_dist:
        t = torch.tensor([peak_rss_gib], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        peak_rss_gib = float(t.item())
    n_found = float(np.mean([len(r) for r in res]))

    # secondary figure: the same K batches through the other retrieval depth (first stage only <-> complete)
    other_qps = None
    if args.with_other_depth and not os.environ.get("SEAL_BENCH_SKIP_OTHER"):
        gc.collect()
        gc.freeze()                                       # keep the timed run's results out of later collections
        searcher.first_stage_only = not args.first_stage_only
        run_batch(0)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t2 = time.perf_counter()
        run_batches(args.warmup, args.steps)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t2 = time.perf_counter() - t2
        if use_dist:
            tt = torch.tensor([t2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t2 = float(tt.item())
        other_qps = args.batch * args.steps * world / t2
        searcher.first_stage_only = args.first_stage_only

    # same-box comparison of an environment switch the product reads per call (SEAL_BENCH_AB="SEAL_SPLIT_GEMM=1,0"): the same K batches
    # with each value in turn, several rounds, after the timed run -- boxes differ by +-15 %, two runs on two boxes say nothing about a 3 %
    # change.  SEAL_BENCH_AB_GARBAGE=1: the legs' results stay with the collector (a host slowed by collections) instead of being frozen.
    ab = None
    if os.environ.get("SEAL_BENCH_AB") and not use_dist:
        var, _, vals = os.environ["SEAL_BENCH_AB"].partition("=")
        vals = vals.split(",") if vals else ["0", "1"]
        ab = {"switch": var, "garbage": bool(os.environ.get("SEAL_BENCH_AB_GARBAGE")), **{v: [] for v in vals}}
        saved = os.environ.get(var)
        for rep in range(int(os.environ.get("SEAL_BENCH_AB_REPS", 3))):
            for val in vals:
                os.environ[var] = val
                if not ab["garbage"]:
                    gc.collect()
                    gc.freeze()                               # (as before the timed run: earlier legs' results out of later collections)
                run_batch(0)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                run_batches(args.warmup, args.steps)
                torch.cuda.synchronize()
                ab[val].append(round(args.batch * args.steps / (time.perf_counter() - t2), 1))
        if saved is None:
            os.environ.pop(var, None)
        else:
            os.environ[var] = saved
        print("[bench] same box, %s (queries/s over %d batches per leg%s): %s" % (var, args.steps, ", results left to the collector" if ab["garbage"] else "",
              "; ".join("%s: %s" % (v, ab[v]) for v in vals)), file=sys.stderr, flush=True)

    world_seen = dist.get_world_size() if use_dist else 1
    try:
        rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version()) if use_dist else None
    except Exception as e:
        rccl_version = "unknown (%r)" % (e,)
    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-phase profile of one extra batch (untimed), also records the index ops for the CPU replay ----
    phases = {}
    gc.collect()
    gc.freeze()                                           # the runs above left long-lived results behind

    def timed(name, fn):
        def wrap(*a, **kw):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = fn(*a, **kw)
            torch.cuda.synchronize()
            phases[name] = phases.get(name, 0.0) + (time.perf_counter() - t) * 1e3
            return out
        return wrap
    orig = (retrieval.fm_index_generate, rk.rescore_keys, rk.compute_unigram_scores, rk.aggregate_evidence_batch, retrieval._count_filter)
    orig_multi, orig_joint = rk.rescore_keys_multi, retrieval.fm_index_generate_joint
    retrieval.fm_index_generate = timed("decode_ms", orig[0])
    retrieval.fm_index_generate._joint_ok = True             # a timer, not another decoder: the searcher keeps its joint loop
    retrieval.fm_index_generate_joint = timed("decode_ms", orig_joint)
    rk.rescore_keys = timed("rescore_ms", orig[1])
    rk.rescore_keys_multi = timed("rescore_ms", orig_multi)      # the searcher's rescorings of a batch: one forward
    rk.compute_unigram_scores = timed("unigram_ms", orig[2])
    rk.aggregate_evidence_batch = timed("aggregate_ms", orig[3])
    retrieval._count_filter = timed("count_filter_ms", orig[4])
    jobs_saved, overlap_saved = searcher.jobs, searcher.overlap
    searcher.overlap = False
    if not os.environ.get("SEAL_BENCH_KEEP_JOBS"):
        searcher.jobs = 1                                  # inline host stages: their time shows up in aggregate_ms
    # launch times: HIP events around every constraint call of this un-overlapped batch, in-kernel counters OFF (they cost the
    # wide launches a few microseconds); the recording pass below runs the same batch once more with the counters on (events
    # off) and supplies the bytes of the very same launches
    # (first the same batch with the product's launch shape -- k_beam_advance ONE launch, bookkeeping and chains together -- events around
    #  every launch: what the chains add to that launch is its duration minus the bookkeeping's, which the next pass times on its own)
    for hd in handles:
        check(lib().fmi_dev_set_option(hd, b"advance_apart", 0))
        check(lib().fmi_dev_enable_timing(hd, 1))
        check(lib().fmi_dev_call_log(hd, 1))
    saved_wrappers = (retrieval.fm_index_generate, retrieval.fm_index_generate_joint, rk.rescore_keys, rk.rescore_keys_multi, rk.compute_unigram_scores,
                      rk.aggregate_evidence_batch, retrieval._count_filter)
    (retrieval.fm_index_generate, retrieval.fm_index_generate_joint, rk.rescore_keys, rk.rescore_keys_multi, rk.compute_unigram_scores,
     rk.aggregate_evidence_batch, retrieval._count_filter) = (orig[0], orig_joint, orig[1], orig_multi, orig[2], orig[3], orig[4])
    run_batch(args.warmup + args.steps)
    (retrieval.fm_index_generate, retrieval.fm_index_generate_joint, rk.rescore_keys, rk.rescore_keys_multi, rk.compute_unigram_scores,
     rk.aggregate_evidence_batch, retrieval._count_filter) = saved_wrappers
    calls_fused = read_call_log(handles[0])
    for hd in handles:
        _ln, _km = ctypes.c_uint64(), ctypes.c_double()
        check(lib().fmi_dev_read_timing(hd, ctypes.byref(_ln), ctypes.byref(_km)))
        check(lib().fmi_dev_set_option(hd, b"advance_apart", 1))
    for hd in handles:
        check(lib().fmi_dev_enable_timing(hd, 1))
        check(lib().fmi_dev_call_log(hd, 1))              # which call each event pair belongs to (prefix length, rows, launch form)
        check(lib().fmi_dev_agg_timing(hd, 1))            # events after every stage of fmi_dev_aggregate (roofline_aggregate)
    if os.environ.get("SEAL_BENCH_PROFILE"):
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        run_batch(args.warmup + args.steps)
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats(os.environ.get("SEAL_BENCH_PROFILE_SORT", "cumulative")).print_stats(int(os.environ.get("SEAL_BENCH_PROFILE_N", "45")))
    else:
        run_batch(args.warmup + args.steps)
    retrieval.fm_index_generate, rk.rescore_keys, rk.compute_unigram_scores, rk.aggregate_evidence_batch, retrieval._count_filter = orig
    rk.rescore_keys_multi, retrieval.fm_index_generate_joint = orig_multi, orig_joint
    import ctypes as C
    calls_timed = read_call_log(handles[0])               # (before fmi_dev_read_timing hands the event pairs back)
    agg_timing = read_agg_timing(handles[0])
    check(lib().fmi_dev_agg_timing(handles[0], 0))
    l2, k2 = C.c_uint64(0), C.c_double(0.0)
    for hd in handles:
        ln, km = C.c_uint64(), C.c_double()
        check(lib().fmi_dev_read_timing(hd, C.byref(ln), C.byref(km)))
        l2.value += ln.value; k2.value += km.value
        check(lib().fmi_dev_enable_timing(hd, 0))
        check(lib().fmi_dev_enable_probe_count(hd, 1))
    # the same batch once more, untimed, RECORDING every index operation with the GPU's answer (parity_check below)
    agg_calls = []

    def recording_aggregate(*a, **kw):
        res = orig[3](*a, **kw)
        agg_calls.append((a, kw, res))
        return res
    rk.aggregate_evidence_batch = recording_aggregate
    trace = []
    index.set_trace(trace)
    run_batch(args.warmup + args.steps)
    index.set_trace(None)
    rk.aggregate_evidence_batch = orig[3]
    calls_counted = read_call_log(handles[0])             # blocks per call, drained call by call in this pass
    check(lib().fmi_dev_call_log(handles[0], 0))
    xstats = [0, 0, 0, 0]
    _p, _l, _k = read_counters(xstats)                     # blocks loaded by the launches of that batch (the replays that
    p2 = ctypes.c_uint64(_p)                               # gpu_allowed_bits issues for the parity check come later)
    for hd in handles:
        check(lib().fmi_dev_enable_probe_count(hd, 0))
    searcher.jobs, searcher.overlap = jobs_saved, overlap_saved

    # one probe = one 128-byte block of the hex wavelet matrix, counted in-kernel (distinct blocks per
    # node): the bytes THIS data structure has to read for the work
    # The launch duration is taken from the un-overlapped batch above (same workload, events around every launch, nothing
    # else on the GPU): in the timed region the previous batch's rescoring / aggregation run on a second stream, and an
    # event pair then also brackets the time its launch waits behind their dispatches -- 71 us there against 34.7 us of
    # execution in the rocprofv3 trace of the very same launches (profiles/r2_kernel_stats.csv).  The timed region's own
    # event figure is kept beside it (`timed_region_event_us`).
    by_call, other_launches = merge_call_logs(calls_timed, calls_counted, calls_fused)
    if by_call:
        # a constraint call = its own launches + the chains k_beam_advance ran for it a model step earlier (by_call adds them); the
        # advance launches without chains (the steps in front of a table call) are the beam loop's bookkeeping, not index work
        k2 = C.c_double(sum(c["us"] for c in by_call) * 1e-3)
        l2 = C.c_uint64(len(by_call))
        k2_upper = sum(c.get("us_upper_bound", c["us"]) for c in by_call) * 1e-3
    n2 = max(1, l2.value)
    achieved = (p2.value * 128.0) / (k2.value * 1e-3) / 1e9 if k2.value > 0 else 0.0
    # SURVEY.md §8(d) prices the same work on the reference-shaped structure (binary wavelet tree, one
    # 64-byte level-probe per node end): counted exactly in-kernel as well, reported beside it
    model_bytes = 2.0 * xstats[3] * 64.0
    model_gbps = model_bytes / (k2.value * 1e-3) / 1e9 if k2.value > 0 else 0.0
    # memory-side traffic comes from a SEPARATE rocprofv3 --pmc FETCH_SIZE pass of this same command over the
    # CURRENT kernel (tools/prof_bench.sh -> profiles/r*_pmc_fetch_size.json, which names the commit it was taken
    # at); a profile of another kernel generation is not used.  Per launch = per constraint call = one k_constrain / k_constrain_table.
    # FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950 (MI355X guide; tools/gather_calib.hip): x 2.
    traffic, traffic_src = cite_traffic(workload_tag)
    nl = n2
    roofline = {"bound": "hbm", "kernel": "k_constrain (one constraint call for the rows of both decodes: the rows' chains -- kept range -> one backward-search step -> class "
                                          "-> root node split -- run by the previous step's k_beam_advance, one wave per row, then k_constrain -- one wave per (row, top digit), "
                                          "the sub-trees level by level by workgroups of 8 waves; the FIRST call of a decode instead k_constrain_table -- the leaf-level nodes "
                                          "of every row from the per-token tables as one evenly cut list -- then k_table_bits; events bracket every launch)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": traffic, "traffic_source": traffic_src, "launches": int(l2.value), "avg_launch_us": round(k2.value * 1e3 / n2, 2),
                "algorithmic_bytes_per_launch": round(p2.value * 128.0 / n2, 1),
                "measured_on": "one batch of the same workload AFTER the timed region, launches alone on the GPU: HIP events around each constraint call with the "
                               "in-kernel counters off, the blocks counted in a second pass over the same batch with the events off; the timed region "
                               "itself runs the product kernels, uninstrumented",
                "survey_8d_model": {"bytes_per_launch": round(model_bytes / nl, 1), "achieved": round(model_gbps, 2),
                                    "frac": round(model_gbps / HBM_PEAK_GBPS, 5),
                                    "note": "binary 16-level wavelet tree, 64 B per level-probe, same symbols emitted"},
                "wave_iterations_per_launch": round(xstats[1] / nl, 1), "lane_pair_utilisation": round(xstats[2] / max(1, 32 * xstats[1]), 3)}

    if by_call:
        roofline["by_call"] = by_call
        roofline["k_beam_advance_bookkeeping_launches"] = other_launches
        roofline["frac_with_chains_as_own_launches"] = round((p2.value * 128.0) / (k2_upper * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if k2_upper > 0 else None
        roofline["widest_call"] = max(by_call, key=lambda c: c["MB"])
        roofline["by_call_note"] = ("one record per constraint call of ONE batch, in launch order: us = HIP events around the call in the un-overlapped timing "
                                    "pass, MB = 128-byte blocks its launches loaded in the counting pass of the same batch; frac = MB / us / 8 TB/s; "
                                    "form: table = k_constrain_table + k_table_bits, row_first = k_constrain_rows + k_constrain, generic = k_constrain, chained = "
                                    "k_constrain started from what the previous step's k_beam_advance left -- the rows' chains, and for rows of <= 64 suffix-array "
                                    "rows the allowed tokens themselves (list mode); chains_us = what the chains add to that launch (its duration as ONE "
                                    "launch minus its bookkeeping's), chains_alone_us = the chains as a launch of their own (upper bound), both included "
                                    "in us / us_upper_bound; chains_MB = their blocks")
        # the same facts as FLAT scalars (a record that keeps only the scalar members of `roofline` still shows the per-call picture)
        wc = roofline["widest_call"]
        roofline.update({"widest_call_frac": wc["frac"], "widest_call_MB": wc["MB"], "widest_call_us": wc["us"], "widest_call_cur_len": wc["cur_len"],
                         "widest_call_form": wc["form"]})
        by_rank = sorted(by_call, key=lambda c: -c["MB"])
        mid = [c for c in by_call if c["form"] != "table" and c["MB"] >= 10.0]      # the wide non-table calls: 3rd..5th token of the two decodes
        if mid:
            mb, us = sum(c["MB"] for c in mid), sum(c["us"] for c in mid)
            roofline.update({"frac_3rd_to_5th_token": round(mb * 1e6 / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4), "MB_3rd_to_5th_token": round(mb, 2),
                             "us_3rd_to_5th_token": round(us, 2), "calls_3rd_to_5th_token": len(mid)})
        roofline["calls_MB_desc"] = ", ".join("%.1f MB @ %.3f" % (c["MB"], c["frac"] or 0.0) for c in by_rank[:5])
    roofline_aggregate = aggregate_roofline(agg_timing, index, workload_tag)

    cpu = parity = None
    if args.no_cpu_baseline and world == 1 and os.environ.get("SEAL_BENCH_SCORE_PARITY") == "1":
        # quick A/B runs of model-side changes (tools/r4_split_gemm.sh): the float half alone, no oracle
        lo_q = (args.warmup + args.steps) * args.batch
        sp = score_parity(searcher, model, index, queries[lo_q:lo_q + args.batch], bias[lo_q:lo_q + args.batch], dev)
        log("score parity vs HF fp32 forward (no oracle leg): beam scores max abs err %.2e / %.2e (body / title), rescoring %.2e, violations %d (tol 1e-4)"
            % (sp["beam_scores_body"]["max_abs_err"], sp["beam_scores_title"]["max_abs_err"], sp["rescore_scores"]["max_abs_err"],
               sp["beam_scores_body"]["violations"] + sp["beam_scores_title"]["violations"] + sp["rescore_scores"]["violations"]))
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = max(1, min(args.cpu_threads or avail, avail))
        t0 = time.perf_counter()
        orc = build_cpu_oracle(index, threads)
        log(f"cpu oracle index (sdsl-style wt_int + rank_support_v, SA/32, ISA/64) built in {time.perf_counter() - t0:.1f}s with {threads} threads")
        # the reference also issues get_count([i]) for the use_top_k_ngrams=5000 best unigrams of every
        # query (keys.py:236-272); here those come from a per-index table, so add them to the replay
        rng = np.random.default_rng(5)
        occ = np.asarray(index.occurring_distinct)
        trace.append(("ranges", [[int(t)] for t in rng.choice(occ, size=5000 * args.batch)], None, None))
        n_real_ops = len(trace) - 1
        stride = max(1, int(args.cpu_locate_sample))
        rep, answers = replay_on_cpu(orc, trace, index.beginnings, threads, locate_stride=stride)
        parity = parity_check(index, trace[:n_real_ops], answers[:n_real_ops], locate_stride=stride)
        if stride > 1:
            parity["locate_and_document_sample"] = f"every {stride}th located row / fully scored document (the host walks ~31 LF steps per row)"
        # the evidence aggregation itself (first stage + full-document scoring on the GPU) against the bit-exact host
        # routines (fmi_first_stage / fmi_full_score) on the same keys: ranked documents, float64 scores, accepted keys
        t0 = time.perf_counter()
        n_docs_cmp = n_bad = 0
        for a, kw, res in agg_calls:
            host = orig[3](*a, **{**kw, "gpu_aggregate": False, "defer": None})
            for (g, _), (w, _) in zip(res, host):
                g = g.result() if hasattr(g, "result") else g
                w = w.result() if hasattr(w, "result") else w
                wl = list(w.items())[:len(g)]
                n_docs_cmp += len(wl)
                for (gd, gi), (wd, wi) in zip(g.items(), wl):
                    same = gd == wd and gi[0] == wi[0] and list(gi[1]) == list(wi[1]) and list(gi[3]) == list(wi[3]) and gi[4][1] == wi[4][1]
                    n_bad += 0 if same else 1
                n_bad += abs(len(g) - len(wl))
        parity["ops"] += len(agg_calls); parity["values_compared"] += n_docs_cmp; parity["mismatches"] += n_bad
        parity["by_kind"]["aggregated_documents_scores_and_keys"] = {"ops": len(agg_calls), "values": n_docs_cmp, "mismatches": n_bad,
                                                                      "against": "fmi_first_stage + fmi_full_score (host float64 routines) on the same keys"}
        log(f"aggregation parity: {n_docs_cmp} ranked documents vs the host routines in {time.perf_counter() - t0:.1f}s, {n_bad} mismatches")
        # ... and, on a sample of the same queries, against the independent scalar model of the reference's aggregate_evidence
        t0 = time.perf_counter()
        ko = aggregation_vs_keys_oracle(orc, agg_calls, n_queries=args.keys_oracle_queries)
        parity["by_kind"]["aggregated_documents_vs_keys_oracle"] = ko
        parity["ops"] += ko["ops"]; parity["values_compared"] += ko["values"]; parity["mismatches"] += ko["mismatches"]
        log(f"aggregation parity: {ko['values']} ranked documents of {ko['queries']} queries vs oracle/keys_oracle.py in {time.perf_counter() - t0:.1f}s "
            f"({ko['rows_located_by_the_oracle']} rows located on the host), {ko['mismatches']} mismatches")
        # the suffix array itself, independently of the builder (the oracle above is fed this index's own BWT / SA samples)
        audit = sa_audit(index)
        audit["text_equals_input_corpus"] = text_is_input
        audit["values"] = audit["adjacent_suffix_pairs"] + audit["lf_rows"]
        audit["mismatches"] += 0 if text_is_input else 1
        parity["by_kind"]["suffix_array_audit"] = audit
        parity["ops"] += 3; parity["values_compared"] += audit["adjacent_suffix_pairs"] + audit["lf_rows"]; parity["mismatches"] += audit["mismatches"]
        log(f"suffix array audit: {audit}")
        # the float half: every recorded hypothesis score of this batch's two decodes, and a sample of its rescoring scores,
        # against HF's own fp32 forward at this geometry (north_star: beam scores within 1e-4)
        t0 = time.perf_counter()
        lo_q = (args.warmup + args.steps) * args.batch
        score_tol = args.bf16_tol if stress else 1e-4
        fp32_model = None
        if stress:
            # the same (bf16-representable) weights, computed in fp32: the FP32 answer  (a fresh module: the bf16 one carries captured graphs)
            with torch.device(dev):
                fp32_model = BartForConditionalGeneration(cfg)
            fp32_model.load_state_dict({k: v.float() for k, v in model.state_dict().items()})
            fp32_model.eval()
        sp = score_parity(searcher, model, index, queries[lo_q:lo_q + args.batch], bias[lo_q:lo_q + args.batch], dev, tol=score_tol, fp32_model=fp32_model)
        if stress:
            for k in ("beam_scores_body", "beam_scores_title", "rescore_scores"):
                sp[k]["arithmetic"] = "bf16 storage, fp32 accumulation, against HF's bf16 forward of the same weights (log-softmax in fp32 on both sides)"
            # the criterion that means something: the recorded scores may be no further from the FP32 forward than 1.5 x what HF's own bf16
            # forward is (floor 0.05) -- instead of a fixed 0.25 against HF's bf16 numbers
            for k in ("beam_scores_body", "beam_scores_title"):
                v = sp[k]["vs_fp32"]
                v["tol_vs_fp32"] = max(0.05, 1.5 * v["hf_lowp_vs_hf_fp32_max_abs_err"])
                v["violation"] = bool(v["max_abs_err_vs_hf_fp32"] > v["tol_vs_fp32"])
                sp[k]["violations"] += int(v["violation"])
            del fp32_model
        beam = {"values": sum(sp[k]["values"] for k in ("beam_scores_body", "beam_scores_title")),
                "max_abs_err": max(sp[k]["max_abs_err"] for k in ("beam_scores_body", "beam_scores_title")), "tol": score_tol,
                "mismatches": sum(sp[k]["violations"] for k in ("beam_scores_body", "beam_scores_title")),
                "body": sp["beam_scores_body"], "title": sp["beam_scores_title"], "against": sp["beam_scores_body"]["against"]}
        parity["by_kind"]["beam_scores"] = beam
        rs = sp["rescore_scores"]
        parity["by_kind"]["rescore_scores"] = {**rs, "mismatches": rs["violations"]}
        parity["ops"] += 3; parity["values_compared"] += beam["values"] + rs["values"]
        parity["mismatches"] += beam["mismatches"] + rs["violations"]
        log(f"score parity vs HF fp32 forward in {time.perf_counter() - t0:.1f}s: beam scores {beam['values']} values, max abs err "
            f"{beam['max_abs_err']:.2e}; rescoring {rs['values']} values, max abs err {rs['max_abs_err']:.2e} (tol {score_tol:g})")
        log(f"parity_check: {parity['ops']} ops, {parity['values_compared']} values, {parity['mismatches']} mismatches")
        # (with a sampled replay the located rows / documents stand for `stride` times as many: their time is scaled up, and said so)
        t_cpu = rep["mask_s"] + rep["ranges_s"] + stride * (rep["locate_s"] + rep["docs_s"])
        cpu = {"value": round(args.batch / t_cpu, 3), "unit": "queries/s (FM-index path only)", "cores": threads, "kind": "port",
               **({"extrapolated": f"locate and document time of every {stride}th row / document, times {stride}"} if stride > 1 else {}),
               "sample": f"FM-index operations of 1 batch of {args.batch} queries (decode-step get_range/get_count from scratch + "
                         f"distinct_count_multi for {rep['rows']} rows, get_count for {rep['sequences']} keys, locate+bisect for "
                         f"{rep['located']} rows, get_doc for {rep['docs']} documents) replayed on the oracle with the reference's call pattern; the model forward is not "
                         f"part of the CPU figure",
               "seconds": {k: round(v, 3) for k, v in rep.items() if k.endswith("_s")}}

    total_q = args.batch * args.steps * world
    out = {
        "metric": (f"queries/sec, 100M-passage synthetic stress FM-index, BART-large bf16 beam={args.beam} batch={args.batch} (complete search)" if stress else
                   "queries/sec, NQ-shaped FM-index, BART-large beam=15 batch=20 (p50 batch latency in extra)"),
        "value": round(total_q / elapsed, 3), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{'configs[4]: 100M-document stress tier, suffix array sorted in slices,' if stress else 'configs[3]: KILT-size' if args.docs >= 30_000_000 else 'configs[1]: NQ-shaped'} synthetic FM-index ({args.docs} passages, {index.size()} symbols{', phrase corpus P=%d' % args.corpus_phrases if args.corpus_phrases else ''}), random-init "
                               f"BART-large {'bf16' if stress else 'fp32'}, beam={args.beam}, batch={args.batch} per GPU, body len 10 + title len<=15, "
                               f"{'first-stage retrieval' if args.first_stage_only else 'first stage + full-document rescoring of 1500 docs/query'}, top-{args.topk}",
                   "index_hbm_gib": round(index.device_bytes() / 2**30, 2), "parallelism": f"query-sharded x{world} ({'torch.distributed backend ' + dist.get_backend() + ' = RCCL, world_size ' + str(dist.get_world_size()) + ', one all_gather of the top-k per timed call' if use_dist else 'one rank, no process group'}), index+model replicated; " + ("decodes of the next two batches enqueued ahead of this batch's rescoring / aggregation; a batch's rescoring forward (library GEMMs) runs beside the next batch's decode steps (hand-written GEMMs only) behind a fence at that decode's library-GEMM prefix, the aggregation overlaps both on the index's stream" if not args.no_overlap else "one batch after the other"), "model_arithmetic": "bf16 storage, fp32 accumulation (BASELINE configs[4])" if stress else "fp32 (as the reference runs BART); linear layers of >= 0.9 GFLOP as one fp16 GEMM over three planes with fp32 accumulation (seal_amd/split_gemm.py; every product of a decode step in sealnn_hgemm_nt), scores within 1e-4 of HF's fp32 forward",
                   "decodes": "body + title of a batch as two loops" if args.no_joint_decode else "body + title of a batch as ONE loop (2 x batch x beams rows per model step, one constraint launch per step)",
                   "query_ngram_keys": "off" if args.no_query_keys else "token 1..3-grams of the query ids (add_query_to_keys=True, the reference's default; "
                                                                                 "spaCy/BART tokenizer absent offline: seal_amd.query_keys.token_ngram_keys)",
                   "not_in_step": (["full-document rescoring (keys.py:366-497)"] if args.first_stage_only else []) +
                                  (["query n-gram keys (add_query_to_keys)"] if args.no_query_keys else [])},
        "p50_batch_latency_ms": round(float(lat[0]), 2) if lat_ms else None,
        "ranks_seen": world_seen, "rccl_version": rccl_version,
        "per_rank_queries_per_s_min": round(min(r[0] for r in per_rank), 2), "per_rank_queries_per_s_max": round(max(r[0] for r in per_rank), 2),
        "per_rank_index_build_s": [round(r[1], 1) for r in per_rank],
        "roofline": roofline,
        "roofline_aggregate": roofline_aggregate,
        "roofline_aggregate_frac": None if not roofline_aggregate else roofline_aggregate.get("frac"),
        "roofline_aggregate_traffic_ratio": None if not roofline_aggregate else roofline_aggregate.get("traffic_ratio"),
        "roofline_aggregate_doc_binning_traffic_ratio": None if not roofline_aggregate else roofline_aggregate.get("doc_binning_traffic_ratio"),
        "roofline_aggregate_total_us": None if not roofline_aggregate else roofline_aggregate.get("total_us"),
        "cpu_baseline": cpu,
        "parity_check": parity,
        "extra": {("complete_search_qps" if args.first_stage_only else "first_stage_only_qps"): None if other_qps is None else round(other_qps, 3),
                  "p50_batch_latency_ms_unpipelined": round(float(lat[0]), 2) if lat_ms else None,
                  "p90_batch_latency_ms_unpipelined": round(float(lat[1]), 2) if lat_ms else None,
                  "latency_batches": len(lat_ms), "latency_note": "single batches after warm-up and the timed region, nothing enqueued ahead, device "
                                                                  "synchronised on both sides; max over ranks of the per-rank median",
                  "warmup_batch_ms": [round(x, 1) for x in step_ms], "docs_returned_per_query": n_found,
                  "peak_host_rss_gib_per_rank_max": round(peak_rss_gib, 2),
                  "timed_region_instrumentation": "none: probe counters and constraint-call event pairs are off during warm-up and the timed call; "
                                                  "roofline figures come from separate un-overlapped passes after it",
                  "decode_step_gemm_algorithms": "sealnn_hgemm_nt for every product of a decode step (split_gemm.HAND_CONFIGS); the library's default picks (hipBLASLt heuristic) for the encoder, the first step and the rescoring forward",
                  "phase_ms_one_batch": {k: round(v, 2) for k, v in phases.items()},
                  "k_constrain_ms_one_batch": round(k2.value, 3), "k_constrain_blocks_one_batch": int(p2.value),
                  "prefix_tables": _prefix_table_stats(index), **({"same_box_ab_qps": ab} if ab else {})},
    }
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and parity["mismatches"]:
        log("PARITY FAILURE: the GPU's answers differ from the CPU oracle's", json.dumps(parity["by_kind"]))
        sys.exit(3)


if __name__ == "__main__":
    main()
