#!/usr/bin/env python
"""Per-call microseconds of the constraint launches of ONE un-overlapped bench batch under several launch-option settings, same box, same
process, same batch: python tools/constrain_ab.py "name=value,name=value" "name=value" ...   ('' = the defaults; options go through
fmi_dev_set_option).  Prints, per variant, the calls' event times (median of --reps passes) and a checksum of what the searcher returned."""
import ctypes, hashlib, json, os, statistics, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from seal_amd import FMIndex
from seal_amd._lib import check, lib
from seal_amd.distributed import pack_topk
from seal_amd.retrieval import SEALSearcher
from transformers import BartConfig, BartForConditionalGeneration

args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 5
docs = int(sys.argv[sys.argv.index("--docs") + 1]) if "--docs" in sys.argv else 21015324
variants = args or [""]
dev = torch.device("cuda", 0)
data, beg, title_len, ids_by_rank = bench.synth_corpus(docs, dev, seed=0, phrases=20000000)
queries, bias = bench.synth_queries(40, data, beg, title_len, ids_by_rank, dev, seed=1)
index = FMIndex(); index.initialize_from_device(data, beg.tolist()); del data
index.labels = None
torch.manual_seed(0)
cfg = BartConfig(); cfg.forced_bos_token_id = None
with torch.device(dev):
    model = BartForConditionalGeneration(cfg)
model.eval()
with torch.no_grad():
    for tok in (cfg.pad_token_id, cfg.bos_token_id, bench.VOCAB - 1):
        model.final_logits_bias[0, tok] = float("-inf")
s = SEALSearcher(index, None, model, detokenize=False, beam=15, batch_size=20, overlap=False)
h = index.handle
DEFAULTS = {}


def run():
    s.logit_bias = bias[20:40]
    return s.batch_search(queries[20:40], k=100)


s.logit_bias = bias[:20]
s.batch_search(queries[:20], k=100)          # warm-up (graphs, tables)
run()
for v in variants:
    opts = dict(kv.split("=") for kv in v.split(",") if kv)
    for k, val in opts.items():
        check(lib().fmi_dev_set_option(h, k.encode(), int(val)))
    run()
    per_call, total = {}, []
    for _ in range(reps):
        check(lib().fmi_dev_set_option(h, b"advance_apart", 0))
        check(lib().fmi_dev_enable_timing(h, 1)); check(lib().fmi_dev_call_log(h, 1))
        res = run()
        torch.cuda.synchronize()
        log = bench.read_call_log(h)
        ln, km = ctypes.c_uint64(), ctypes.c_double()
        check(lib().fmi_dev_read_timing(h, ctypes.byref(ln), ctypes.byref(km)))
        check(lib().fmi_dev_enable_timing(h, 0)); check(lib().fmi_dev_call_log(h, 0))
        t = 0.0
        for c in log:
            if c["form"].startswith("advance"):
                continue
            per_call.setdefault((c["cur_len"], c["form"]), []).append(c["us"]); t += c["us"]
        total.append(t)
    digest = hashlib.sha256(pack_topk(res, 100).numpy().tobytes()).hexdigest()[:16]
    print("variant %-40s own launches of the constraint calls: %.1f us per batch (median of %d; min %.1f)  top-100 sha %s" %
          (v or "(defaults)", statistics.median(total), reps, min(total), digest))
    print("   " + "  ".join("%d:%s %.1f" % (cl, form[:5], statistics.median(us)) for (cl, form), us in per_call.items()))
    for k in opts:                               # back to the built-in choice
        check(lib().fmi_dev_set_option(h, k.encode(), -1))
