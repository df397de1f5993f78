"""End-to-end numerics of the split GEMM at BART-large geometry, on the CPU (no GPU needed): token log-probabilities of a teacher-forced
decode through HF's own forward with (a) fp32 linears, (b) every nn.Linear / lm_head product replaced by the emulated three-plane fp16
product of seal_amd/split_gemm.py (products summed in fp32), against (c) the float64 forward.  Prints the worst and rms error of the summed
log-probabilities of 10-token hypotheses -- the quantity north_star bounds by 1e-4."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from transformers import BartConfig, BartForConditionalGeneration
from seal_amd.split_gemm import SplitLinears

torch.manual_seed(0)
cfg = BartConfig()
cfg.forced_bos_token_id = None
model = BartForConditionalGeneration(cfg).eval()
n_seq, T, S = int(os.environ.get("N_SEQ", 12)), 10, 16
g = torch.Generator().manual_seed(1)
enc = torch.randint(4, 50000, (n_seq, S), generator=g)
dec = torch.randint(4, 50000, (n_seq, T), generator=g)
dec[:, 0] = 2
tgt = torch.randint(4, 50000, (n_seq, T), generator=g)


def logprob_sums(m, dtype=torch.float32):
    with torch.no_grad():
        logits = m(input_ids=enc, decoder_input_ids=dec).logits.to(dtype)
        lp = torch.log_softmax(logits.float() if dtype == torch.float32 else logits, dim=-1)
        return lp.gather(-1, tgt[..., None])[..., 0].cumsum(-1).double()          # running sums = recorded hypothesis scores


t = time.time()
a = logprob_sums(model)
print(f"fp32 forward {time.time() - t:.1f}s", flush=True)
split = SplitLinears()
orig = F.linear


def split_linear(x, w, b=None):
    if x.dtype != torch.float32 or w.dim() != 2 or w.shape[1] % 4:
        return orig(x, w, b)
    y = split(x.reshape(-1, x.shape[-1]), w, b)
    return y.view(*x.shape[:-1], w.shape[0])


F.linear = split_linear
torch.nn.functional.linear = split_linear
try:
    t = time.time()
    b = logprob_sums(model)
    print(f"split forward {time.time() - t:.1f}s ({len(split._by_weight)} weights split)", flush=True)
finally:
    F.linear = orig
    torch.nn.functional.linear = orig
t = time.time()
ref = logprob_sums(model.double(), torch.float64)
print(f"float64 forward {time.time() - t:.1f}s", flush=True)
for name, v in (("fp32 linears", a), ("split linears", b)):
    e = (v - ref).abs()
    print(f"{name:14s} vs float64: max abs err of the running log-prob sums {e.max().item():.3e}, rms {e.pow(2).mean().sqrt().item():.3e}")
print(f"split vs fp32: max {((a - b).abs().max().item()):.3e}   (tolerance of north_star: 1e-4; |score| up to {ref.abs().max().item():.1f})")
