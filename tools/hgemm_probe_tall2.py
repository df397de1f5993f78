"""where a K step of the tall tile goes: the full kernel, the kernel without LDS-DMA after its prologue (reads + MFMAs + barriers only), and without
reads / MFMAs (LDS-DMA + waits + barriers only); W rotated over copies larger than the 256 MB memory-side cache, as a decode step sees its weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.hgemm_probe import gtime, dev
from seal_amd._lib import check, lib
from seal_amd import split_gemm
L = lib()

def run(a, ws, cfg, out, i=[0]):
    w = ws[i[0] % len(ws)]; i[0] += 1
    M, K = a.shape
    check(L.sealnn_hgemm_nt(torch.cuda.current_stream(dev).cuda_stream, a.data_ptr(), w.data_ptr(), out.data_ptr(), M, w.shape[0], K, w.shape[0], cfg))

torch.manual_seed(0)
for M in (600, 300):
    for name, N, K, sl in [("d x d", 1024, 3072, 8), ("qkv", 3072, 3072, 2 if M == 600 else 4), ("fc1", 4096, 3072, 2 if M == 600 else 4), ("fc2", 1024, 12288, 8 if M == 600 else 16)]:
        a = torch.randn(M, K, device=dev).half()
        copies = max(2, int(640e6 / (N * K * 2)))
        ws = [torch.randn(N, K, device=dev).half() for _ in range(copies)]
        out = torch.empty(16, M, N, dtype=torch.float32, device=dev)
        cur = split_gemm.hand_config(M, N, K)
        base = 6 | (3 << 8) | (1 << 12) | (sl << 16)
        ts = [gtime(lambda: run(a, ws, c, out)) for c in (cur, base, base | (1 << 30), base | (2 << 30))]
        one = [gtime(lambda: run(a, ws[:1], c, out)) for c in (cur, base)]
        print(f"M={M:4d} {name:6s} x{sl}: today {ts[0]:6.1f}  tall {ts[1]:6.1f}  tall without DMA {ts[2]:6.1f}  tall without reads/MFMA {ts[3]:6.1f}   (one resident W: today {one[0]:.1f} tall {one[1]:.1f})", flush=True)
        del ws
