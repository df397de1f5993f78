import time, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def gtime(fn, n=200):
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
tot = {}
for M in (300, 600):
    tot[M] = 0
    for name, N, K, cnt in [("qkv", 3072, 1024, 12), ("proj", 1024, 1024, 36), ("fc1", 4096, 1024, 12), ("fc2", 1024, 4096, 12), ("lm_head", 50265, 1024, 1)]:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        t = gtime(lambda: F.linear(x, w, b))
        tot[M] += t * cnt
        print(M, name, round(t, 1), "us")
print({k: round(v / 1e3, 3) for k, v in tot.items()}, "ms of GEMM per decode step")
