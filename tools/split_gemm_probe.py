"""First thing to run in round 4 (one GPU call, ~1 min): is the split GEMM of seal_amd/split_gemm.py worth switching on?
  1. does torch.addmm(fp32 bias, fp16, fp16, out_dtype=float32) run on this ROCm build (hipBLASLt fp16 in / fp32 out)?
  2. time, inside hipGraphs, F.linear fp32 against SplitLinear (split kernel + one fp16 GEMM over 3K) for the shapes of a decode step
     (600 / 300 / 40 rows) and of the rescoring forward (3200 rows);
  3. error of both against float64;  4. what the matrix cores do with fp16 subnormals.
Prints one JSON object per shape and a per-forward summary."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from seal_amd.split_gemm import SplitLinear, overflowed, split_planes_reference
dev = torch.device("cuda:0")


def gtime(fn, n=100):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n // 20):
            g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


try:
    a = torch.randn(8, 16, device=dev).half(); b = torch.randn(4, 16, device=dev).half()
    y = torch.addmm(torch.zeros(4, device=dev), a, b.t(), alpha=0.5, out_dtype=torch.float32)
    print("addmm fp16 -> fp32:", y.dtype, "max err", float((y.double() - 0.5 * a.double() @ b.double().t()).abs().max()))
except Exception as e:
    print("addmm fp16 -> fp32 NOT available:", type(e).__name__, str(e)[:300])
    sys.exit(2)
# subnormals: 2^-20 * 2^10 = 2^-10 if the operand survives, 0 if it is flushed
tiny = torch.full((32, 32), 2.0 ** -20, device=dev).half(); big = torch.full((32, 32), 2.0 ** 10 / 32, device=dev).half()
print("fp16 subnormal operand in the matrix cores ->", float(torch.mm(tiny, big.t(), out_dtype=torch.float32)[0, 0]), "(expected", 2.0 ** -10, ")")
res = {}
g = torch.Generator().manual_seed(0)
for M in (600, 300, 40, 3200):
    for name, N, K, cnt in (("qkv", 3072, 1024, 12), ("proj", 1024, 1024, 36), ("fc1", 4096, 1024, 12), ("fc2", 1024, 4096, 12), ("lm", 50265, 1024, 1)):
        x = torch.randn(M, K, generator=g).to(dev); x[:, :4] *= 30
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev); b = torch.randn(N, generator=g).to(dev)
        lin = SplitLinear(w, b)
        r = {"M": M, "N": N, "K": K, "per_forward": cnt}
        r["fp32_us"] = round(gtime(lambda: F.linear(x, w, b)), 1)
        r["split_us"] = round(gtime(lambda: lin(x)), 1)
        planes = split_planes_reference(x)                       # GEMM alone, planes given
        r["split_gemm_only_us"] = round(gtime(lambda: torch.addmm(lin.bias, planes, lin.wt, alpha=lin.alpha, out_dtype=torch.float32)), 1)
        r["fp16_TFs"] = round(2 * M * N * 3 * K / r["split_gemm_only_us"] / 1e6, 1)
        r["fp32_TFs"] = round(2 * M * N * K / r["fp32_us"] / 1e6, 1)
        if N <= 4096:
            ref = x.double() @ w.double().t() + b.double()
            r["rms_err_fp32"] = float((F.linear(x, w, b).double() - ref).pow(2).mean().sqrt())
            r["rms_err_split"] = float((lin(x).double() - ref).pow(2).mean().sqrt())
        res[f"{name}_{M}"] = r
        print(json.dumps({f"{name}_{M}": r}), flush=True)
print("activations out of fp16 range:", overflowed(dev))
for M in (600, 300, 40, 3200):
    d = sum(v["fp32_us"] * v["per_forward"] for k, v in res.items() if v["M"] == M)
    s = sum(v["split_us"] * v["per_forward"] for k, v in res.items() if v["M"] == M)
    o = sum(v["split_gemm_only_us"] * v["per_forward"] for k, v in res.items() if v["M"] == M)
    print(f"M={M}: linear layers of one forward  fp32 {d / 1e3:.3f} ms   split {s / 1e3:.3f} ms   (its GEMMs alone {o / 1e3:.3f} ms)")
