import os, sys, time, torch
dev = torch.device("cuda:0")
def bench(M, N, K, reps=20):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    for _ in range(3): torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return dt * 1e3, 2 * M * N * K / dt / 1e12
lib = sys.argv[1] if len(sys.argv) > 1 else "default"
if lib != "default": torch.backends.cuda.preferred_blas_library(lib)
print("blas:", torch.backends.cuda.preferred_blas_library(), "tunable:", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"))
tot = 0
for name, (M, N, K), cnt in [("qkv", (300, 3072, 1024), 12), ("out", (300, 1024, 1024), 24), ("cq", (300, 1024, 1024), 12), ("fc1", (300, 4096, 1024), 12), ("fc2", (300, 1024, 4096), 12), ("lm_head", (300, 50265, 1024), 1)]:
    ms, tf = bench(M, N, K); tot += ms * cnt
    print(f"{name:8s} M{M} N{N} K{K}: {ms:.3f} ms  {tf:.1f} TF/s  x{cnt}")
print("sum per step ms", round(tot, 2))
