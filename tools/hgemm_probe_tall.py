"""the tall tile of sealnn_hgemm_nt (320 x 128 / 320 x 64, 8 waves: tile codes 5 / 6) against the 4-wave tiles that serve a decode step today, product
by product at 600 and 300 rows: exact on one-hot operands first, then us per call inside a graph.
python tools/hgemm_probe_tall.py > profiles/r6_hgemm_probe_tall.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.hgemm_probe import gtime, hgemm, dev
from seal_amd import split_gemm

torch.manual_seed(0)
bad = 0
print("# correctness of the tall tiles: max |C - ref| / max |ref| on random operands, one-hot operands exact")
for (M, N, K) in [(600, 1024, 3072), (300, 3072, 3072), (37, 200, 192), (640, 1024, 12288), (600, 4096, 3072), (321, 50265, 3072), (1000, 130, 256)]:
    a = torch.randn(M, K, device=dev).half()
    w = torch.randn(N, K, device=dev).half()
    ref = torch.mm(a.float(), w.float().t())
    a1 = torch.zeros(M, K, device=dev, dtype=torch.float16)
    a1[torch.arange(M, device=dev), (torch.arange(M, device=dev) * 7 + 3) % K] = 1.0
    ref1 = torch.mm(a1.float(), w.float().t())
    for tile, stages in ((5, 2), (6, 2), (6, 3), (7, 2), (7, 3)):
        for slices in (1, 2, 3, 4):
            if (K // 64) % slices:
                continue
            cfg = tile | (stages << 8) | (1 << 12) | (slices << 16)
            c = hgemm(a, w, cfg).sum(0)
            c1 = hgemm(a1, w, cfg).sum(0)
            torch.cuda.synchronize()
            e = float((c - ref).abs().max() / ref.abs().max())
            ok = e < 1e-5 and torch.equal(c1, ref1)
            bad += not ok
            print(f"  M={M} N={N} K={K} tile {tile} stages {stages} slices {slices}: {e:.2e} one-hot {'exact' if torch.equal(c1, ref1) else 'WRONG'}{'' if ok else '   <-- FAIL'}", flush=True)
print("# %d failures" % bad)

print("# us per call (graph of 20; W rotated over copies of > 600 MB in all, as a decode step meets its weights: not in any cache): today's configuration vs the tall tiles")
from seal_amd._lib import check, lib
L = lib()
def run(a, ws, cfg, out, i=[0]):
    w = ws[i[0] % len(ws)]; i[0] += 1
    check(L.sealnn_hgemm_nt(torch.cuda.current_stream(dev).cuda_stream, a.data_ptr(), w.data_ptr(), out.data_ptr(), a.shape[0], w.shape[0], a.shape[1], w.shape[0], cfg))
for M in (600, 300):
    for name, N, K, slice_opts in [("d x d", 1024, 3072, (2, 3, 4, 6, 8)), ("qkv", 3072, 3072, (1, 2, 3, 4, 6)), ("fc1", 4096, 3072, (1, 2, 3, 4, 6)),
                                   ("fc2", 1024, 12288, (4, 6, 8, 12, 16)), ("lm_head", 50265, 3072, (1,))]:
        if name == "lm_head" and M != 600:
            continue
        a = torch.randn(M, K, device=dev).half()
        ws = [torch.randn(N, K, device=dev).half() for _ in range(max(2, int(640e6 / (N * K * 2))))]
        n = 100 if N > 10000 else 200
        cur = split_gemm.hand_config(M, N, K)
        out = torch.empty(16, M, N, dtype=torch.float32, device=dev)
        t_cur = gtime(lambda: run(a, ws, cur, out), n=n)
        res = []
        for tile, stages in ((5, 2), (6, 3), (7, 2), (7, 3)):
            for slices in slice_opts:
                if (K // 64) % slices:
                    continue
                cfg = tile | (stages << 8) | (1 << 12) | (slices << 16)
                res.append((gtime(lambda: run(a, ws, cfg, out), n=n), tile, stages, slices))
        res.sort()
        fmt = lambda r: f"{r[0]:.1f} (tile {r[1]} stages {r[2]} x{r[3]})"
        print(f"M={M:4d} {name:7s} N={N:5d} K={K:5d}: today {t_cur:6.1f} (config {cur:#x})   tall: {', '.join(fmt(r) for r in res[:6])}", flush=True)
        del ws
