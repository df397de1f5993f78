"""SEAL_TUNED_GEMMS=tune:<path>: does a decode of a shape that is not in the shipped file get tuned during the capture warm-up, and
do the picks land in <path>?  (BART-large geometry, 3 queries x 5 beams.)"""
import glob, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
out = sys.argv[1]
os.environ["SEAL_TUNED_GEMMS"] = "tune:" + out
import torch
from transformers import BartConfig, BartForConditionalGeneration
from seal_amd.bart_decoder import BartStepDecoder
from seal_amd import tuned_gemm
dev = torch.device("cuda:0")
cfg = BartConfig()
with torch.device(dev):
    model = BartForConditionalGeneration(cfg).eval()
dec = BartStepDecoder(model)
ids = torch.randint(4, 1000, (3, 12), device=dev)
mask = torch.ones_like(ids)
with torch.no_grad():
    enc = dec.encode(ids, mask)
    t = time.perf_counter()
    dec.start(enc, mask, 5, 10)
    torch.cuda.synchronize()
    print("mode", tuned_gemm.setup(), "start+capture", round(time.perf_counter() - t, 1), "s")
    logits = dec.step(torch.full((15,), 2, device=dev))
    torch.cuda.synchronize()
import torch.cuda.tunable as tn
print("filename", tn.get_filename(), "results", len(tn.get_results()))
for r in tn.get_results()[:40]:
    if "_15_" in r[1]:
        print(r)
print("files", glob.glob(os.path.dirname(out) + "/*tuned*"))
