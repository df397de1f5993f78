import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seal_amd import FMIndex
from seal_amd.keys import aggregate_evidence
AGG = json.load(open("tests/golden/ref_aggregate_evidence.json"))["cases"]
unhex = float.fromhex
nbad = 0
for cn, case in enumerate(AGG):
    kw = case["kwargs"]
    if kw.get("sort_by_length") or kw.get("sort_by_freq"):
        continue
    print("case", cn, flush=True)
    ix = FMIndex(); ix.initialize(case["docs"])
    keys = [(list(k), unhex(s)) for k, s in case["keys"]]
    us = None if case["unigram_scores"] is None else [unhex(x) for x in case["unigram_scores"]]
    ix._agg_debug = []
    res, _ = aggregate_evidence(keys, unigram_scores=us, index=ix, **kw)
    fs = ix._agg_debug[0][0] if ix._agg_debug else None
    ix._agg_debug = None
    os.environ["SEAL_HOST_AGGREGATE"] = "1"
    ranked, _ = aggregate_evidence(keys, unigram_scores=us, index=ix, first_stage_only=True, **kw)
    del os.environ["SEAL_HOST_AGGREGATE"]
    got = [(int(d), info[0]) for d, info in res.items()]
    want = [(w["doc"], unhex(w["score"])) for w in case["results"]]
    ok_fs = fs is None or (fs[0].tolist() == list(ranked.keys()) and fs[1].tolist() == [i[0] for i in ranked.values()])
    if got != want or not ok_fs:
        nbad += 1
        print("CASE", cn, {k: v for k, v in kw.items() if k in ("allow_overlaps", "single_key", "beta", "n_docs_complete_score", "max_occurrences_1")})
        print("  first stage ok:", ok_fs)
        if not ok_fs:
            print("   gpu :", list(zip(fs[0].tolist(), fs[1].tolist()))[:40])
            print("   host:", [(d, i[0]) for d, i in ranked.items()][:40])
        print("  full got :", got[:40])
        print("  full want:", want[:40])
print("bad cases:", nbad)
