"""host-side profile of the per-query key scoring of aggregate_evidence (keys.py:207-309) on NQ-like sizes, no GPU"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seal_amd import keys as rk

V = 50265
rng = np.random.default_rng(0)


class FakeIndex:
    beginnings = [0, 2_879_038_741]
    occurring_distinct = list(range(4, V))

    def __len__(self):
        return self.beginnings[-1]

    def get_range_batch(self, seqs):
        n = len(seqs)
        lens = np.asarray([len(s) for s in seqs])
        cnt = np.maximum(1, (3e7 / (40.0 ** (lens - 1))).astype(np.int64))
        lo = rng.integers(0, 2_000_000_000, size=n).astype(np.uint64)
        return lo, lo + cnt.astype(np.uint64)


def make_query():
    keys = []
    for b in range(15):
        base = rng.integers(4, V, size=9).tolist()
        for n in range(1, 10):
            keys.append((base[:n], -float(rng.random() * 3 + 0.3 * n)))
    for b in range(15):
        base = [2] + rng.integers(4, V, size=5).tolist() + [49314]
        keys.append((base, -float(rng.random() * 6)))
    us = np.log(rng.dirichlet(np.full(V, 0.05)) + 1e-30)
    return keys, us


ix = FakeIndex()
queries = [make_query() for _ in range(20)]
params = dict(max_occurrences_1=1500, n_docs_complete_score=1500, alpha=2.0, beta=0.8, add_best_unigrams_to_ngrams=True,
              use_top_k_unigrams=5000, smoothing=5.0)


def run():
    out = []
    for keys, us in queries:
        g = rk._aggregate_steps(keys, us, ix, **params)
        out.append(next(g))
    return out


run()
t0 = time.perf_counter()
for _ in range(3):
    reqs = run()
print("score_split per batch of 20: %.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

# ---- the C++ route (fmi_agg_score_pack): flatten + score + pack, host only ----
import ctypes
from itertools import chain
from seal_amd._lib import check, lib
ulo = np.zeros(V, dtype=np.int64); uhi = ulo + rng.integers(1, 10_000_000, size=V)


def cpp():
    nq = len(queries)
    key_lists = [[k for k, _ in keys] for keys, _ in queries]
    lens = [len(k) for keys in key_lists for k in keys]
    lm = [s for keys, _ in queries for _, s in keys]
    q_off = np.zeros(nq + 1, dtype=np.int64); q_off[1:] = np.cumsum([len(k) for k in key_lists])
    tok_off = np.zeros(len(lens) + 1, dtype=np.int64); np.cumsum(lens, out=tok_off[1:])
    toks = np.fromiter(chain.from_iterable(chain.from_iterable(key_lists)), dtype=np.int64, count=int(tok_off[-1]))
    lm = np.asarray(lm, dtype=np.float64)
    lo, hi = ix.get_range_batch([k for keys in key_lists for k in keys])
    ptrs = (ctypes.c_void_p * nq)(*[us.ctypes.data for _, us in queries])
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    plan = ctypes.c_void_p()
    check(lib().fmi_agg_score_pack(nq, p(q_off), p(tok_off), p(toks), p(lm), p(lo), p(hi), ptrs, V, p(ulo), p(uhi), V, float(ix.beginnings[-1]),
                                   2.0, 0.0, 5.0, 1, 1, 5000, 1500, 10_000_000, ix.beginnings[-1] + 1, ctypes.byref(plan)))
    lib().fmi_agg_plan_free(plan)


cpp()
t0 = time.perf_counter()
for _ in range(5):
    cpp()
print("C++ score+pack per batch of 20 (incl. python flattening): %.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
