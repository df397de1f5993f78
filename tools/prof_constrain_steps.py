#!/usr/bin/env python
"""Per-call numbers of the constraint launches of ONE bench batch: rows, cur_len, blocks probed, microseconds, how many rows
are dead / empty / narrow / wide (allowed tokens).  python tools/prof_constrain_steps.py [--docs N]"""
import ctypes, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from seal_amd import FMIndex, beam_search
from seal_amd._lib import check, lib
from seal_amd.retrieval import SEALSearcher
from transformers import BartConfig, BartForConditionalGeneration

docs = int(sys.argv[sys.argv.index("--docs") + 1]) if "--docs" in sys.argv else 21015324
dev = torch.device("cuda", 0)
data, beg, title_len, ids_by_rank = bench.synth_corpus(docs, dev, seed=0, phrases=20000000)
queries, bias = bench.synth_queries(40, data, beg, title_len, ids_by_rank, dev, seed=1)
index = FMIndex(); index.initialize_from_device(data, beg.tolist()); del data
torch.manual_seed(0)
cfg = BartConfig(); cfg.forced_bos_token_id = None
with torch.device(dev):
    model = BartForConditionalGeneration(cfg)
model.eval()
with torch.no_grad():
    for tok in (cfg.pad_token_id, cfg.bos_token_id, bench.VOCAB - 1):
        model.final_logits_bias[0, tok] = float("-inf")
s = SEALSearcher(index, None, model, detokenize=False, beam=15, batch_size=20)
h = index.handle
real = beam_search.fused_topk_groups
log = []
def wrapped(procs, batches, input_ids, logits, beam_scores, K, parent_rows=None, tag=0):
    torch.cuda.synchronize()
    check(lib().fmi_dev_enable_probe_count(h, 1)); check(lib().fmi_dev_enable_timing(h, 1))
    pr, ln, km = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double()
    check(lib().fmi_dev_read_probe_count(h, ctypes.byref(pr))); check(lib().fmi_dev_read_timing(h, ctypes.byref(ln), ctypes.byref(km)))
    out = real(procs, batches, input_ids, logits, beam_scores, K, parent_rows=parent_rows, tag=tag)
    torch.cuda.synchronize()
    check(lib().fmi_dev_read_probe_count(h, ctypes.byref(pr))); check(lib().fmi_dev_read_timing(h, ctypes.byref(ln), ctypes.byref(km)))
    if input_ids.shape[1] >= 2:
        rows = input_ids.shape[0]
        bits = torch.zeros(rows, (bench.VOCAB + 31) // 32, dtype=torch.int32, device=dev)
        a = 0
        pops = []
        for p, b in zip(procs, batches):
            ids = input_ids[a:a + b * K].contiguous()
            ff = p.force_decoding_from or []
            ff_arr = (ctypes.c_int64 * max(len(ff), 1))(*ff)
            bb = torch.zeros(b * K, (bench.VOCAB + 31) // 32, dtype=torch.int32, device=dev)
            check(lib().fmi_dev_allowed_bits(h, torch.cuda.current_stream(dev).cuda_stream, b * K, ids.shape[1], ids.data_ptr(), bb.data_ptr(), bench.VOCAB,
                                             bench.SHIFT, p.pad_token_id, p.eos_token_id, ff_arr, len(ff), 0, 0))
            torch.cuda.synchronize()
            x = bb.cpu().numpy().view(np.uint32)
            pops.append(np.unpackbits(x.view(np.uint8), axis=1).sum(1))
            a += b * K
        pop = np.concatenate(pops)
        log.append(dict(cur_len=int(input_ids.shape[1]), rows=int(rows), blocks=int(pr.value), MB=round(pr.value * 128 / 1e6, 1), us_with_counters=round(km.value * 1e3, 1),
                        rows_le1=int((pop <= 1).sum()), rows_2_64=int(((pop > 1) & (pop <= 64)).sum()), rows_65_1024=int(((pop > 64) & (pop <= 1024)).sum()),
                        rows_gt1024=int((pop > 1024).sum()), max_allowed=int(pop.max())))
        check(lib().fmi_dev_read_probe_count(h, ctypes.byref(pr))); check(lib().fmi_dev_read_timing(h, ctypes.byref(ln), ctypes.byref(km)))
    return out
s.logit_bias = bias[:20]
s.batch_search(queries[:20], k=100)          # warm-up (graphs)
beam_search.fused_topk_groups = wrapped
s.logit_bias = bias[20:40]
s.overlap = False
s.batch_search(queries[20:40], k=100)
for e in log:
    print(json.dumps(e))
