import sys, time
sys.path.insert(0, "/root/repo")
import torch
from transformers import BartConfig, BartForConditionalGeneration
from seal_amd.bart_decoder import BartStepDecoder
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = BartConfig(); cfg.forced_bos_token_id = None
with torch.device(dev):
    model = BartForConditionalGeneration(cfg).eval()
for B in (20, 40, 80):
    dec = BartStepDecoder(model)
    ids = torch.randint(4, 50000, (B, 30), device=dev); am = torch.ones_like(ids)
    enc = dec.encode(ids, am); dec.start(enc, am, 15, 15)
    toks = torch.full((B * 15,), 2, device=dev)
    for _ in range(3): dec.step(toks)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dec.step(toks)
    torch.cuda.synchronize(); print("B", B, "rows", B * 15, "ms/step", (time.perf_counter() - t0) * 100)
