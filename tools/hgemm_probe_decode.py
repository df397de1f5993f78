"""the three products of a decode step that round 5 left to hipBLASLt (qkv at 600 rows, fc1, lm_head), sealnn_hgemm_nt against the library's
fp16 GEMM, us per call inside a graph: python tools/hgemm_probe_decode.py > profiles/r6_hgemm_probe_decode.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.hgemm_probe import gtime, hgemm, dev

torch.manual_seed(0)
for M in (600, 300, 40):
    for name, N, K, slice_opts in [("qkv", 3072, 3072, (1, 2, 4)), ("fc1", 4096, 3072, (1,)), ("lm_head", 50265, 3072, (1,))]:
        a = torch.randn(M, K, device=dev).half()
        w = torch.randn(N, K, device=dev).half()
        wt = w.t()
        t_lib = gtime(lambda: torch.mm(a, wt, out_dtype=torch.float32), n=100 if N > 10000 else 200)
        res = []
        for tile in (1, 2, 3, 4, 129, 130, 131, 132):
            for stages, kg in ((2, 1), (3, 1), (2, 2), (2, 4)):
                if kg > {1: 1, 2: 4, 3: 2, 4: 2}[tile & 127]:
                    continue
                for slices in slice_opts:
                    if (K // 64) % slices or (K // 64 // slices) % kg:
                        continue
                    cfg = tile | (stages << 8) | (kg << 12) | (slices << 16)
                    out = torch.empty(slices, M, N, dtype=torch.float32, device=dev)
                    res.append((gtime(lambda: hgemm(a, w, cfg, out), n=100 if N > 10000 else 200), tile, stages, kg, slices))
        res.sort()
        fmt = lambda r: f"{r[0]:.1f} (tile {r[1]} stages {r[2]} kgroups {r[3]}{'' if r[4] == 1 else ' x%d slices' % r[4]})"
        print(f"M={M:5d} {name:7s} N={N:5d} K={K:5d}: library {t_lib:6.1f}   best: {', '.join(fmt(r) for r in res[:4])}", flush=True)
