"""Experiment: fp32 GEMMs of one BART-large decode step (M = 300 rows) with the library's default algorithm
vs PyTorch TunableOp's pick (stock rocBLAS / hipBLASLt solutions, benchmarked per shape)."""
import os, sys, time, json
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
shapes = [("qkv", 300, 3072, 1024), ("proj1024", 300, 1024, 1024), ("fc1", 300, 4096, 1024), ("fc2", 300, 1024, 4096), ("lm_head", 300, 50265, 1024),
          ("resc_qkv", 4608, 3072, 1024), ("resc_proj", 4608, 1024, 1024), ("resc_fc1", 4608, 4096, 1024), ("resc_fc2", 4608, 1024, 4096), ("resc_lm", 4608, 50265, 1024)]
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
res = {}
ws = {}
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    ws[name] = (x, w, b)
    res[name] = {"default_us": round(bench(lambda: F.linear(x, w, b)), 1), "gflop": 2*M*N*K/1e9}
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True)
try:
    tn.set_max_tuning_duration(200); tn.set_max_tuning_iterations(30)
except Exception as e:
    print("tunable knobs:", e)
t0 = time.perf_counter()
for name, M, N, K in shapes:
    x, w, b = ws[name]
    F.linear(x, w, b); torch.cuda.synchronize()
print("tuning took", round(time.perf_counter()-t0, 1), "s", file=sys.stderr)
tn.tuning_enable(False)
for name, M, N, K in shapes:
    x, w, b = ws[name]
    res[name]["tuned_us"] = round(bench(lambda: F.linear(x, w, b)), 1)
    r = res[name]; r["default_TF"] = round(r["gflop"]/r["default_us"]/1e3*1e3, 1); r["tuned_TF"] = round(r["gflop"]/r["tuned_us"]/1e3*1e3, 1)
print(json.dumps(res, indent=1))
try:
    print(tn.get_results()[:12])
except Exception as e:
    print(e)
