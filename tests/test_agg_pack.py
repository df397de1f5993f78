"""Host side of the GPU evidence aggregation: ``fmi_agg_pack`` (seal_amd/csrc/fmi_agg_pack.cpp) packs the scored keys
of a chunk of queries into the blob the kernels read.  Checked here against plain python: query-local token ids,
distinct-token sets (repetition(), reference keys.py:186-191), the heap order of keys (keys.py:431), the trie
(keys.py:377-384: walking it finds exactly the keys), occurrence slots cut to max_hits, sparse unigram scores."""
import ctypes
import struct

import numpy as np
import pytest

from seal_amd._lib import check, lib

HDR_FIELDS = ["magic", "bytes", "nq", "n_keys", "n_rare", "total_occ", "vocab", "max_key_len", "max_u", "max_q_keys", "n_uni",
              "n_trie_slots", "n_tok", "o_q_key_off", "o_q_rare_off", "o_rare_key", "o_rare_occ_off", "o_key_lo", "o_key_len", "o_key_q",
              "o_key_rank", "o_key_score", "o_kset_off", "o_kset_ids", "o_q_tok_off", "o_tok_list", "o_q_trie_off", "o_trie", "o_uni_flat",
              "o_uni_score"]


def _hash(node, tok):
    h = ((node * 0x9E3779B1) ^ (tok * 0x85EBCA77)) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0xC2B2AE3D) & 0xFFFFFFFF
    h ^= h >> 13
    return h


def _pack(queries, max_hits, index_size, type_scores, vocab):
    q_off = np.zeros(len(queries) + 1, dtype=np.int64)
    toks, tok_off, score, rare, lo, hi = [], [0], [], [], [], []
    for i, keys in enumerate(queries):
        q_off[i + 1] = q_off[i] + len(keys)
        for k, s, r, a, b in keys:
            toks += list(k)
            tok_off.append(len(toks))
            score.append(s); rare.append(r); lo.append(a); hi.append(b)
    arr = lambda x, dt: np.ascontiguousarray(np.asarray(x if len(x) else [0], dtype=dt))
    tok_off, toks, score = arr(tok_off, np.int64), arr(toks, np.int64), arr(score, np.float64)
    rare, lo, hi = arr(rare, np.uint8), arr(lo, np.uint64), arr(hi, np.uint64)
    ts = [None if t is None else np.ascontiguousarray(t, dtype=np.float64) for t in type_scores]
    tp = (ctypes.c_void_p * len(queries))(*[None if t is None else t.ctypes.data for t in ts])
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    plan = ctypes.c_void_p()
    check(lib().fmi_agg_pack(len(queries), p(q_off), p(tok_off), p(toks), p(score), p(rare), p(lo), p(hi), max_hits, index_size, tp, vocab,
                             ctypes.byref(plan)))
    n = ctypes.c_uint64()
    ptr = lib().fmi_agg_plan_blob(plan, ctypes.byref(n))
    blob = ctypes.string_at(ptr, n.value)
    assert lib().fmi_agg_plan_occurrences(plan) == struct.unpack_from("<Q", blob, 8 * HDR_FIELDS.index("total_occ"))[0]
    lib().fmi_agg_plan_free(plan)
    H = dict(zip(HDR_FIELDS, struct.unpack_from("<%dQ" % len(HDR_FIELDS), blob, 0)))
    assert H["bytes"] == len(blob) and blob[:8] == b"SFMIAGG1"
    return H, blob


def test_plan_blob_matches_python():
    rng = np.random.default_rng(0)
    vocab = 50
    queries, types = [], []
    for q in range(3):
        seen, keys = set(), []
        for _ in range(40):
            k = tuple(rng.integers(3, 12, size=int(rng.integers(1, 6))).tolist())
            if k in seen:
                continue
            seen.add(k)
            a = int(rng.integers(0, 900))
            keys.append((k, float(rng.integers(1, 5)), int(rng.random() < 0.6), a, a + int(rng.integers(0, 30))))   # tied scores on purpose
        keys.sort(key=lambda x: -x[1])
        queries.append(keys)
        t = np.zeros(vocab)
        t[rng.integers(0, vocab, size=7)] = rng.random(7) + 0.1
        types.append(t if q != 1 else None)
    H, blob = _pack(queries, max_hits=10, index_size=905, type_scores=types, vocab=vocab)
    u32 = lambda o, n: np.frombuffer(blob, dtype=np.uint32, count=n, offset=H[o])
    u64 = lambda o, n: np.frombuffer(blob, dtype=np.uint64, count=n, offset=H[o])
    f64 = lambda o, n: np.frombuffer(blob, dtype=np.float64, count=n, offset=H[o])
    nq, nk = 3, sum(len(k) for k in queries)
    assert (H["nq"], H["n_keys"], H["vocab"]) == (nq, nk, vocab)
    q_key, q_rare, q_tok, q_trie = u32("o_q_key_off", nq + 1), u32("o_q_rare_off", nq + 1), u32("o_q_tok_off", nq + 1), u32("o_q_trie_off", nq + 1)
    rare_key, occ = u32("o_rare_key", H["n_rare"]), u64("o_rare_occ_off", H["n_rare"] + 1)
    key_len, key_q, key_rank = u32("o_key_len", nk), u32("o_key_q", nk), u32("o_key_rank", nk)
    key_lo, key_score = u64("o_key_lo", nk), f64("o_key_score", nk)
    kset_off = u32("o_kset_off", nk + 1)
    kset = u32("o_kset_ids", int(kset_off[-1]))
    tok_list = u32("o_tok_list", H["n_tok"])
    trie = u32("o_trie", 4 * H["n_trie_slots"]).reshape(-1, 4)
    flat = [k for keys in queries for k in keys]
    assert key_len.tolist() == [len(k[0]) for k in flat] and key_score.tolist() == [k[1] for k in flat]
    assert H["max_key_len"] == max(len(k[0]) for k in flat) and H["max_q_keys"] == max(len(k) for k in queries)
    r = 0
    for q, keys in enumerate(queries):
        k0 = int(q_key[q])
        assert q_key[q + 1] - k0 == len(keys) and (key_q[k0:k0 + len(keys)] == q).all()
        local = sorted({t for k in keys for t in k[0]})
        assert tok_list[q_tok[q]:q_tok[q + 1]].tolist() == local
        for j, (k, s, rare, a, b) in enumerate(keys):
            ids = kset[kset_off[k0 + j]:kset_off[k0 + j + 1]].tolist()
            assert sorted(local[i] for i in ids) == sorted(set(k)) and len(ids) == len(set(k))
            if rare:
                assert rare_key[r] == k0 + j and key_lo[k0 + j] == a
                assert occ[r + 1] - occ[r] == min(max(min(b, 905) - a, 0), 10)
                r += 1
        assert q_rare[q + 1] == r
        # heap order of keys.py:431: (-score, token tuple)
        want = sorted(range(len(keys)), key=lambda j: (-keys[j][1], keys[j][0]))
        assert [int(np.flatnonzero(key_rank[k0:k0 + len(keys)] == x)[0]) for x in range(len(keys))] == want
        # the trie: walking any token sequence finds exactly the keys that are prefixes of it
        base, cap = int(q_trie[q]), int(q_trie[q + 1] - q_trie[q])
        assert cap & (cap - 1) == 0
        keyset = {k[0]: k0 + j for j, k in enumerate(keys)}

        def walk(seq):
            node, found = 0, []
            for d, t in enumerate(seq):
                i = _hash(node, t) & (cap - 1)
                while True:
                    s = trie[base + i]
                    if s[0] == 0xFFFFFFFF:
                        return found
                    if s[0] == node and s[1] == t:
                        break
                    i = (i + 1) & (cap - 1)
                node = int(s[2])
                if s[3] != 0xFFFFFFFF:
                    found.append((d + 1, int(s[3])))
            return found
        for _ in range(300):
            seq = tuple(rng.integers(3, 12, size=7).tolist())
            assert walk(seq) == [(n, keyset[seq[:n]]) for n in range(1, 8) if seq[:n] in keyset]
    assert H["total_occ"] == occ[-1]
    uni = {int(f): s for f, s in zip(u64("o_uni_flat", H["n_uni"]), f64("o_uni_score", H["n_uni"]))}
    assert uni == {q * vocab + t: float(types[q][t]) for q in range(nq) if types[q] is not None for t in np.flatnonzero(types[q])}


def test_plan_rejects_what_the_kernels_cannot_take():
    import pytest
    from seal_amd._lib import SealFMError
    with pytest.raises(SealFMError):       # a key without a positive score never reaches the trie (keys.py:378)
        _pack([[((3, 4), 0.0, 1, 0, 5)]], 10, 100, [None], 1)
    with pytest.raises(SealFMError):       # window lengths are 8-bit in the position sort keys
        _pack([[(tuple(range(3, 3 + 300)), 1.0, 1, 0, 5)]], 10, 100, [None], 1)


# ---------------------------------------------------------------------------
# fmi_agg_score_pack: the key scoring of aggregate_evidence (reference keys.py:207-309) in C++ against the python that
# is itself held to the reference's vectors (seal_amd.keys._aggregate_steps, tests/test_reference_golden.py)
# ---------------------------------------------------------------------------
def _score_pack(index, jobs, kw):
    from seal_amd.keys import _unigram_ranges
    nq = len(jobs)
    q_off = np.zeros(nq + 1, dtype=np.int64)
    toks, tok_off, lm, ptrs, alive = [], [0], [], [], []
    vocab = 1
    for i, (keys, us) in enumerate(jobs):
        q_off[i + 1] = q_off[i] + len(keys)
        for k, s in keys:
            toks += list(k)
            tok_off.append(len(toks))
            lm.append(s)
        if us is not None:
            us = np.ascontiguousarray(us, dtype=np.float64)
            alive.append(us); ptrs.append(us.ctypes.data); vocab = max(vocab, len(us))
        else:
            ptrs.append(None)
    flat = [list(k) for keys, _ in jobs for k, _ in keys]
    lo, hi = index.get_range_batch(flat)
    arr = lambda x, dt: np.ascontiguousarray(np.asarray(x if len(x) else [0], dtype=dt))
    tok_off, toks, lm = arr(tok_off, np.int64), arr(toks, np.int64), arr(lm, np.float64)
    lo, hi = arr(lo, np.uint64), arr(hi, np.uint64)
    ulo, uhi = _unigram_ranges(index)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    tp = (ctypes.c_void_p * nq)(*ptrs)
    plan = ctypes.c_void_p()
    check(lib().fmi_agg_score_pack(nq, p(q_off), p(tok_off), p(toks), p(lm), p(lo), p(hi), tp if alive else None, vocab, p(ulo), p(uhi), len(ulo),
                                   float(index.beginnings[-1]), float(kw.get("alpha", 2.0)), float(kw.get("length_penalty", 0.0)),
                                   float(kw.get("smoothing", 5.0)), int(kw.get("use_fm_index_frequency", True)),
                                   int(kw.get("add_best_unigrams_to_ngrams", False)), int(kw.get("use_top_k_unigrams", 1000)),
                                   int(kw.get("max_occurrences_1", 1500)), int(kw.get("max_occurrences_2", 10_000_000)), int(index.orc.size()),
                                   ctypes.byref(plan)))
    out = []
    for q in range(nq):
        n = int(lib().fmi_agg_plan_ngrams(plan, q, None, None, None))
        src, sc, rare = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.float64), np.zeros(max(n, 1), np.uint8)
        lib().fmi_agg_plan_ngrams(plan, q, p(src), p(sc), p(rare))
        keys = jobs[q][0]
        out.append([((tuple(keys[s][0]) if s >= 0 else (-s - 1,)), v, bool(r)) for s, v, r in zip(src[:n].tolist(), sc[:n].tolist(), rare[:n].tolist())])
    n = ctypes.c_uint64()
    blob = ctypes.string_at(lib().fmi_agg_plan_blob(plan, ctypes.byref(n)), n.value)
    lib().fmi_agg_plan_free(plan)
    return out, blob


@pytest.mark.parametrize("seed,kw", [
    (0, dict()),
    (1, dict(add_best_unigrams_to_ngrams=True, use_top_k_unigrams=30)),
    (2, dict(add_best_unigrams_to_ngrams=True, use_top_k_unigrams=5000, max_occurrences_1=5, alpha=1.5, smoothing=0.5)),
    (3, dict(use_fm_index_frequency=False, add_best_unigrams_to_ngrams=True, use_top_k_unigrams=12, length_penalty=0.2)),
    (4, dict(add_best_unigrams_to_ngrams=True, use_top_k_unigrams=0)),
    (5, dict(max_occurrences_2=8, length_penalty=0.1, alpha=1.0)),
])
def test_cpp_key_scoring_equals_the_python_scoring(seed, kw):
    import pytest  # noqa: F401
    from oracle.seal_oracle import OracleFMIndex
    from seal_amd.keys import _aggregate_steps
    from tests.helpers import OracleBatchIndex, make_docs, synthetic_keys
    vocab = 60
    rng = np.random.default_rng(seed)
    docs = make_docs(seed, 150, vocab, min_len=6, max_len=20, title_sep=7)
    orc = OracleFMIndex()
    orc.initialize(docs)
    index = OracleBatchIndex(orc)
    jobs = []
    for q in range(4):
        keys = synthetic_keys(rng, docs, vocab, n_keys=50, with_titles=True)
        keys += [(k, s - 0.5) for k, s in keys[:3]]                         # repeated n-grams: dict semantics (value replaced, place kept)
        keys += [([int(rng.integers(4, vocab))], -float(int(rng.integers(1, 4)))) for _ in range(6)]      # single tokens, tied scores
        us = None if q == 3 else np.round(-rng.random(vocab) * 6 - 0.01, 1 if q == 1 else 6)             # q == 1: ties at the top-k boundary
        jobs.append((keys, us))
    got, blob = _score_pack(index, jobs, kw)
    params = dict(max_occurrences_1=1500, n_docs_complete_score=500, alpha=2.0, beta=0.8, smoothing=5.0, use_top_k_unigrams=1000)
    params.update(kw)
    H = dict(zip(HDR_FIELDS, struct.unpack_from("<%dQ" % len(HDR_FIELDS), blob, 0)))
    n_table = 0
    for q, (keys, us) in enumerate(jobs):
        g = _aggregate_steps(keys, None if us is None else us.tolist(), index, **params)
        try:
            req = next(g)
            rare, all_ngrams = req[4]["rare"], req[4]["all_ngrams"]
            g.close()
        except StopIteration:                                   # no rare key at all: the python returns without asking for rows
            rare, all_ngrams = {}, None
        if all_ngrams is not None:
            assert [(k, v) for k, v, _ in got[q]] == list(all_ngrams.items()), (q, kw)      # order and float64 values exactly
            assert [k for k, _, r in got[q] if r] == list(rare.keys())
            n_table += sum(1 for _, v, _ in got[q] if v > 0.0)
        else:
            assert not any(r for _, _, r in got[q])
            n_table += sum(1 for _, v, _ in got[q] if v > 0.0)
    assert H["n_keys"] == n_table and H["nq"] == 4
