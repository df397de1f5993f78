"""the PAIRS form of the decode step's products (hi / lo pairs per 32 columns: four tiles per K step instead of six) against today's three-block
configurations, W rotated through memory: us per call inside a graph, every tile and slice count.
python tools/hgemm_probe_pairs.py > profiles/r6_hgemm_probe_pairs.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.hgemm_probe import gtime, dev
from seal_amd._lib import check, lib
from seal_amd import split_gemm
L = lib()
def run(a, ws, cfg, out, i=[0]):
    w = ws[i[0] % len(ws)]; i[0] += 1
    check(L.sealnn_hgemm_nt(torch.cuda.current_stream(dev).cuda_stream, a.data_ptr(), w.data_ptr(), out.data_ptr(), a.shape[0], w.shape[0], a.shape[1], w.shape[0], cfg))
torch.manual_seed(0)
for M in ([int(a) for a in sys.argv[1:]] or [600, 300]):
    for name, N, K, slice_opts in [("d x d", 1024, 1024, (1, 2, 4, 8, 16)), ("qkv", 3072, 1024, (1, 2, 4, 8)), ("fc1", 4096, 1024, (1, 2, 4, 8)),
                                   ("fc2", 1024, 4096, (2, 4, 8, 16, 32)), ("lm_head", 50265, 1024, (1,))]:
        n = 100 if N > 10000 else 200
        out = torch.empty(16, M, N, dtype=torch.float32, device=dev)
        a3 = torch.randn(M, 3 * K, device=dev).half()
        ws3 = [torch.randn(N, 3 * K, device=dev).half() for _ in range(max(2, int(640e6 / (N * K * 6))))]
        cur = split_gemm.hand_config(M, N, 3 * K) or (2 | (2 << 8) | (1 << 12) | (1 << 16))
        t_cur = gtime(lambda: run(a3, ws3, cur, out), n=n)
        del ws3
        a2 = torch.randn(M, 2 * K, device=dev).half()
        ws2 = [torch.randn(N, 2 * K, device=dev).half() for _ in range(max(2, int(640e6 / (N * K * 4))))]
        res = []
        for tile, stage_opts, kgs in ((1, (2,), (1,)), (129, (2, 3), (1,)), (2, (2, 3), (1, 2, 4)), (3, (2, 3), (1, 2)), (4, (2, 3), (1, 2)), (132, (2, 3), (1,)), (5, (2,), (1,)), (6, (3,), (1,)), (7, (3,), (1,))):
            if name == "lm_head" and tile in (2, 3):
                continue
            for stages in stage_opts:
                for kg in kgs:
                    if kg > 1 and stages != 2:
                        continue
                    for slices in slice_opts:
                        if (2 * K // 64) % slices or (2 * K // 64 // slices) % kg:
                            continue
                        cfg = tile | (stages << 8) | (kg << 12) | (slices << 16) | split_gemm.PAIRS_BIT
                        res.append((gtime(lambda: run(a2, ws2, cfg, out), n=n), tile, stages, kg, slices))
        res.sort()
        fmt = lambda r: f"{r[0]:.1f} (tile {r[1]} stages {r[2]} kgroups {r[3]} x{r[4]})"
        print(f"M={M:4d} {name:7s} N={N:5d} K={K:5d}: three blocks today {t_cur:6.1f} (config {cur:#x})   pairs: {', '.join(fmt(r) for r in res[:6])}", flush=True)
        del ws2
