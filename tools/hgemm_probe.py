"""sealnn_hgemm_nt (seal_amd/csrc/hgemm_kernels.hip) against torch on an MI355X: exact on one-hot operands (a transposed / permuted fragment
layout cannot pass), within fp32-accumulation noise on random ones, for every tile / pipelining / split-K configuration; then microseconds
per call next to the library's fp16 GEMM on the decode step's shapes.   usage: python tools/hgemm_probe.py > profiles/r5_hgemm_probe.txt"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seal_amd._lib import check, lib

dev = torch.device("cuda:0")
L = lib()


def hgemm(a, w, config=0, out=None):
    M, K = a.shape
    N = w.shape[0]
    slices = max(1, config >> 16)
    c = out if out is not None else torch.empty(slices, M, N, dtype=torch.float32, device=dev)
    check(L.sealnn_hgemm_nt(torch.cuda.current_stream(dev).cuda_stream, a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, N, config))
    return c


def gtime(fn, n=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n // 20):
            g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def main():
    torch.manual_seed(0)
    bad = 0
    print("# correctness: max |C - ref| / max |ref| (one-hot operands must be exact)")
    for (M, N, K) in [(64, 64, 64), (128, 128, 128), (600, 1024, 3072), (300, 3072, 3072), (37, 200, 192), (640, 1024, 12288), (600, 4096, 3072)]:
        a = torch.randn(M, K, device=dev).half()
        w = torch.randn(N, K, device=dev).half()
        ref = torch.mm(a.float(), w.float().t())
        # one-hot rows of A: C[i][j] = W[j][i mod K] exactly
        a1 = torch.zeros(M, K, device=dev, dtype=torch.float16)
        a1[torch.arange(M, device=dev), (torch.arange(M, device=dev) * 7 + 3) % K] = 1.0
        ref1 = torch.mm(a1.float(), w.float().t())
        for tile in (1, 2, 3, 4):
            for stages, kg in ((1, 1), (2, 1), (3, 1), (2, 2), (2, 4)):
                if kg > {1: 1, 2: 4, 3: 2, 4: 2}[tile]:
                    continue
                for slices in (1, 2, 3):
                    if (K // 64) % slices or (K // 64 // slices) % kg:
                        continue
                    cfg = tile | (stages << 8) | (kg << 12) | (slices << 16)
                    c = hgemm(a, w, cfg).sum(0)
                    c1 = hgemm(a1, w, cfg).sum(0)
                    torch.cuda.synchronize()
                    e = float((c - ref).abs().max() / ref.abs().max())
                    e1 = float((c1 - ref1).abs().max())
                    ok = e < 2e-3 and e1 == 0.0
                    bad += 0 if ok else 1
                    if not ok or (tile == 2 and stages == 2 and slices == 1):
                        print(f"M={M} N={N} K={K} tile={tile} stages={stages} kgroups={kg} slices={slices}: random {e:.2e}  one-hot {e1:.2e}  {'ok' if ok else 'WRONG'}")
    print(f"# {bad} wrong configurations")
    print("# us per call: library fp16 GEMM (torch.mm out fp32) vs sealnn_hgemm_nt per configuration (tile/pipelined/slices)")
    for M in (600, 300, 3200):
        for name, N, K in [("proj", 1024, 3072), ("qkv", 3072, 3072), ("fc1", 4096, 3072), ("fc2", 1024, 12288)]:
            a = torch.randn(M, K, device=dev).half()
            w = torch.randn(N, K, device=dev).half()
            wt = w.t()
            t_lib = gtime(lambda: torch.mm(a, wt, out_dtype=torch.float32))
            res = []
            for tile in (1, 2, 3, 4):
                for stages, kg in ((2, 1), (3, 1), (2, 2), (2, 4)):
                    if kg > {1: 1, 2: 4, 3: 2, 4: 2}[tile]:
                        continue
                    for slices in (1, 2, 4):
                        if (K // 64) % slices or (K // 64 // slices) % kg:
                            continue
                        cfg = tile | (stages << 8) | (kg << 12) | (slices << 16)
                        out = torch.empty(slices, M, N, dtype=torch.float32, device=dev)
                        res.append((gtime(lambda: hgemm(a, w, cfg, out)), tile, stages, kg, slices))
            res.sort()
            auto = gtime(lambda: hgemm(a, w, 0))
            fmt = lambda r: f"{r[0]:.1f} (tile {r[1]} stages {r[2]} kgroups {r[3]}{'' if r[4] == 1 else ' x%d slices' % r[4]})"
            print(f"M={M:5d} {name:5s} N={N:5d} K={K:5d}: library {t_lib:6.1f}   auto {auto:6.1f}   best: {', '.join(fmt(r) for r in res[:3])}   | one slab: {', '.join(fmt(r) for r in [r for r in res if r[4] == 1][:3])}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
