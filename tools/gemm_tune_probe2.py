"""fp32 GEMMs of the joint decode step (M = 600 / 300) and of the rescoring forward (M ~ 3200): library default vs PyTorch TunableOp's
pick, timed inside a hipGraph (20 calls per replay).  Writes the tuned picks to gpurun_out/tunableop_results0.csv."""
import os, sys, time, json
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "tunableop_results.csv")
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
shapes = []
for M in (600, 300, 3200):
    shapes += [(f"qkv_{M}", M, 3072, 1024, 12), (f"proj_{M}", M, 1024, 1024, 36), (f"fc1_{M}", M, 4096, 1024, 12), (f"fc2_{M}", M, 1024, 4096, 12),
               (f"lm_{M}", M, 50265, 1024, 1)]
def gtime(fn, n=100):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n // 20): g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6
ws, res = {}, {}
for name, M, N, K, cnt in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    ws[name] = (x, w, b)
    res[name] = {"default_us": round(gtime(lambda: F.linear(x, w, b)), 1), "gflop": 2 * M * N * K / 1e9, "per_step": cnt}
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True)
try:
    tn.set_max_tuning_duration(300); tn.set_max_tuning_iterations(50)
except Exception as e:
    print("tunable knobs:", e, file=sys.stderr)
t0 = time.perf_counter()
for name, M, N, K, cnt in shapes:
    x, w, b = ws[name]
    F.linear(x, w, b); torch.cuda.synchronize()
print("tuning took", round(time.perf_counter() - t0, 1), "s", file=sys.stderr)
tn.tuning_enable(False)
for name, M, N, K, cnt in shapes:
    x, w, b = ws[name]
    r = res[name]
    r["tuned_us"] = round(gtime(lambda: F.linear(x, w, b)), 1)
    r["default_TF"] = round(r["gflop"] / r["default_us"] * 1e3 / 1e3, 1); r["tuned_TF"] = round(r["gflop"] / r["tuned_us"] * 1e3 / 1e3, 1)
for M in (600, 300, 3200):
    d = sum(res[n]["default_us"] * res[n]["per_step"] for n in res if n.endswith(f"_{M}"))
    t = sum(res[n]["tuned_us"] * res[n]["per_step"] for n in res if n.endswith(f"_{M}"))
    print(f"M={M}: GEMMs of one forward  default {d / 1e3:.3f} ms   tuned {t / 1e3:.3f} ms")
print(json.dumps(res))
try:
    tn.write_file()
except Exception as e:
    print("write_file:", e, file=sys.stderr)
