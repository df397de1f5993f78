"""Experiment: host cost per eager F.linear call (hipBLASLt vs rocBLAS backend) and GPU time at rescoring shapes."""
import time, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def host_cost(M, N, K, n=400):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    for _ in range(5): F.linear(x, w, b)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): F.linear(x, w, b)
    t_host = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t) / n * 1e6
    return round(t_host, 1), round(t_all, 1)
for lib in ("hipblaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable", e); continue
    print(lib, "tiny 8x1024x1024 (host us, total us):", host_cost(8, 1024, 1024), " resc 2816x1024x1024:", host_cost(2816, 1024, 1024, 100),
          " 2816x3072x1024:", host_cost(2816, 3072, 1024, 100), " 2816x4096x1024:", host_cost(2816, 4096, 1024, 100), " 2816x1024x4096:", host_cost(2816, 1024, 4096, 100),
          " 300x1024x1024:", host_cost(300, 1024, 1024, 200))
