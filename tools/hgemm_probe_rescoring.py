"""the products of the rescoring forward (a prefix tree of ~3 100 .. 3 400 nodes per batch: seal_amd/keys.py) and of a batch's encoder (40 inputs x 32 tokens):
the library's fp16 GEMM (what they run on) against sealnn_hgemm_nt's tiles, one slab, W rotated through memory.
python tools/hgemm_probe_rescoring.py > profiles/r6_hgemm_probe_rescoring.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.hgemm_probe import gtime, dev
from seal_amd._lib import check, lib
L = lib()

def run(a, ws, cfg, out, i=[0]):
    w = ws[i[0] % len(ws)]; i[0] += 1
    check(L.sealnn_hgemm_nt(torch.cuda.current_stream(dev).cuda_stream, a.data_ptr(), w.data_ptr(), out.data_ptr(), a.shape[0], w.shape[0], a.shape[1], w.shape[0], cfg))

def run_lib(a, ws, out, i=[0]):
    w = ws[i[0] % len(ws)]; i[0] += 1
    torch.mm(a, w.t(), out=out) if out.dtype == a.dtype else torch.mm(a, w.t(), out_dtype=torch.float32)

torch.manual_seed(0)
for M in (3328, 1280):
    for name, N, K in [("d x d", 1024, 3072), ("ckv", 2048, 3072), ("qkv", 3072, 3072), ("fc1", 4096, 3072), ("fc2", 1024, 12288)]:
        a = torch.randn(M, K, device=dev).half()
        ws = [torch.randn(N, K, device=dev).half() for _ in range(max(2, int(640e6 / (N * K * 2))))]
        out = torch.empty(4, M, N, dtype=torch.float32, device=dev)
        t_lib = gtime(lambda: run_lib(a, ws, out[0]), n=100)
        res = []
        for tile, stage_opts in ((1, (2, 3)), (129, (2, 3)), (3, (2, 3)), (4, (2, 3)), (5, (2,)), (6, (3,)), (7, (3,))):
            for stages in stage_opts:
                for slices in (1, 2):
                    if (K // 64) % slices:
                        continue
                    cfg = tile | (stages << 8) | (1 << 12) | (slices << 16)
                    res.append((gtime(lambda: run(a, ws, cfg, out), n=100), tile, stages, slices))
        # the PAIRS form of the same product: [M, 2K/3] x [N, 2K/3]^T pair planes, three products per K step
        K2 = K // 3 * 2
        a2 = torch.randn(M, K2, device=dev).half()
        ws2 = [torch.randn(N, K2, device=dev).half() for _ in range(max(2, int(640e6 / (N * K2 * 2))))]
        for tile, stage_opts in ((1, (2, 3)), (129, (2, 3)), (3, (2, 3)), (4, (2, 3)), (132, (2, 3)), (5, (2,)), (6, (3,)), (7, (3,))):
            for stages in stage_opts:
                for slices in (1, 2):
                    if (K2 // 64) % slices:
                        continue
                    cfg = tile | (stages << 8) | (1 << 12) | (slices << 16) | (1 << 29)
                    res.append((gtime(lambda: run(a2, ws2, cfg, out), n=100), 1000 + tile, stages, slices))
        del ws2
        res.sort()
        fmt = lambda r: f"{r[0]:.1f} ({'pairs ' if r[1] >= 1000 else ''}tile {r[1] % 1000} stages {r[2]} x{r[3]})"
        gf = 2.0 * M * N * K / 1e9
        print(f"M={M:4d} {name:6s} N={N:5d} K={K:5d} ({gf:5.1f} GF): library {t_lib:6.1f} us   hand: {', '.join(fmt(r) for r in res[:5])}", flush=True)
        del ws
