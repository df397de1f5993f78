import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from seal_amd import FMIndex
from seal_amd.beam_search import IndexBasedLogitsProcessor, _inf_nan_remove, _history_to_hypotheses
from seal_amd.bart_decoder import BartStepDecoder
from transformers import BartConfig, BartForConditionalGeneration
dev = torch.device("cuda:0")
data, beg, tl, ids_by_rank = bench.synth_corpus(int(os.environ.get("DOCS", 2000000)), dev)
queries, bias = bench.synth_queries(20, data, beg, tl, ids_by_rank, dev)
index = FMIndex(); index.initialize_from_device(data, beg.tolist()); del data
torch.manual_seed(0)
cfg = BartConfig(); cfg.forced_bos_token_id = None
with torch.device(dev):
    model = BartForConditionalGeneration(cfg).eval()
dec = BartStepDecoder(model); dec.logit_bias = bias
from seal_amd import keys as rk
ids = rk._pad_batch(queries, 1, dev); am = (ids != 1).long()
B, K, T = 20, 15, 10
proc = IndexBasedLogitsProcessor(index, K, pad_token_id=1, eos_token_id=2)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    tm = {}
    def add(k, t0): tm[k] = tm.get(k, 0) + (sync() - t0) * 1e3
    t0 = sync(); enc = dec.encode(ids, am); add("encode", t0)
    t0 = sync(); dec.start(enc, am, K, T); add("start", t0)
    R = B * K
    input_ids = torch.full((R, 1), 2, dtype=torch.long, device=dev)
    beam_scores = torch.zeros(B, K, device=dev); beam_scores[:, 1:] = -1e9; beam_scores = beam_scores.view(R)
    row_base = (torch.arange(B, device=dev) * K).unsqueeze(1)
    steps = []
    while True:
        t0 = sync(); logits = dec.step(input_ids[:, -1]); add("step", t0)
        t0 = sync(); logp = torch.log_softmax(logits.float(), -1); processed = _inf_nan_remove(logp); add("logsoftmax+infnan", t0)
        V = processed.shape[-1]
        t0 = sync(); unc = processed + beam_scores[:, None]; add("add_beam", t0)
        t0 = sync(); con = proc(input_ids, processed) + beam_scores[:, None]; add("mask", t0)
        t0 = sync(); _, flat = torch.topk(con.view(B, K * V), 2 * K, dim=1); add("topk", t0)
        t0 = sync()
        ns = unc.view(B, K * V).gather(-1, flat); ni = flat // V; nt = flat % V; src = row_base + ni
        steps.append((input_ids[src], nt, ns))
        keep = nt != 2
        order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)[:, :K]
        beam_scores = ns.gather(1, order).view(R); bt = nt.gather(1, order).view(R); bi = src.gather(1, order).view(R)
        input_ids = torch.cat([input_ids[bi], bt.unsqueeze(-1)], -1); add("select", t0)
        t0 = sync(); dec.reorder(bi); add("reorder", t0)
        if input_ids.shape[-1] >= T: break
    t0 = sync(); out = _history_to_hypotheses(steps, (input_ids, beam_scores), B, K, 0.0); add("history_to_python", t0)
    print(rep, {k: round(v, 2) for k, v in tm.items()}, "total", round(sum(tm.values()), 1), file=sys.stderr)
