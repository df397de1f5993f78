"""The decode step's linear layers, shape by shape, against their floors (measure first).
For every (M, N, K) of a BART-large decode step (M = 600 joint / 300 title-only rows; 3200 = the rescoring forward): the library's fp32
GEMM, the library's fp16 GEMM over the three split planes (K' = 3K, fp32 accumulate: what seal_amd/split_gemm.py issues), and -- when
libsealfm exports it -- the hand-written gfx950 kernel (sealnn_split_gemm).  floor = max(flops / peak, weight bytes / 8 TB/s, 5 us)
with peak = 2.5 PFLOP/s (fp16 MFMA, dense) for the 3K products and 157 TFLOP/s for fp32.
usage: python tools/gemm_shape_table.py > profiles/r5_gemm_shape_table.txt"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")


def gtime(fn, n=200):
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


def main():
    shapes = [("qkv", 3072, 1024, 12), ("proj 1024x1024 (so / cq / co)", 1024, 1024, 36), ("fc1", 4096, 1024, 12), ("fc2", 1024, 4096, 12),
              ("lm_head", 50265, 1024, 1)]
    try:
        from seal_amd.hip_gemm import split_gemm_raw          # the hand-written kernel, when built
    except Exception:
        split_gemm_raw = None
    print("# us per call; TF = useful fp32-equivalent TFLOP/s (2 M N K / t); x floor = t / max(3 * 2MNK / 2.5e15, 6 N K / 8e12, 5 us)")
    print("%-32s %5s %6s %5s | %9s %7s | %9s %7s %8s | %9s %7s %8s" % ("layer", "M", "N", "K", "fp32 us", "TF", "split us", "TF", "x floor", "hand us", "TF", "x floor"))
    tot = {}
    for M in (600, 300, 3200):
        tot[M] = [0.0, 0.0, 0.0]
        for name, N, K, cnt in shapes:
            x = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev)
            xp = torch.randn(M, 3 * K, device=dev).half()
            wp = torch.randn(N, 3 * K, device=dev).half()
            wt = wp.t()
            t32 = gtime(lambda: torch.mm(x, w.t()))
            t16 = gtime(lambda: torch.mm(xp, wt, out_dtype=torch.float32))
            flops = 2.0 * M * N * K
            floor = max(3 * flops / 2.5e15, 6.0 * N * K / 8e12, 5e-6) * 1e6
            th = None
            if split_gemm_raw is not None:
                try:
                    th = gtime(lambda: split_gemm_raw(xp, wp))
                except Exception as e:
                    th = None
            print("%-32s %5d %6d %5d | %9.1f %7.1f | %9.1f %7.1f %8.1f | %9s %7s %8s" % (
                name, M, N, K, t32, flops / t32 / 1e6, t16, flops / t16 / 1e6, t16 / floor,
                "%.1f" % th if th else "-", "%.1f" % (flops / th / 1e6) if th else "-", "%.1f" % (th / floor) if th else "-"))
            tot[M][0] += t32 * cnt; tot[M][1] += min(t32, t16) * cnt; tot[M][2] += (min(t32, t16, th) if th else min(t32, t16)) * cnt
    for M, (a, b, c) in tot.items():
        print("# M = %d: one decoder pass of 12 layers + lm_head: fp32 %.2f ms, best of fp32 / split per layer %.2f ms, with the hand-written kernel %.2f ms" % (M, a / 1e3, b / 1e3, c / 1e3))


if __name__ == "__main__":
    main()
