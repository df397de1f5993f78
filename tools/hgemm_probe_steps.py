"""where the cycles of a tall-tile workgroup go, step by step (s_memtime in waves 0 and 7: top of the step, after the vmcnt wait, after the barrier)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.hgemm_probe import dev
from seal_amd._lib import check, lib
L = lib()
torch.manual_seed(0)
for tile, M, name, N, K, sl in [(7, 600, "qkv", 3072, 3072, 4), (6, 600, "qkv", 3072, 3072, 2)]:
    a = torch.randn(M, K, device=dev).half()
    ws = [torch.randn(N, K, device=dev).half() for _ in range(max(2, int(640e6 / (N * K * 2))))]
    out = torch.zeros(16, M, N, dtype=torch.float32, device=dev)
    cfg = tile | (3 << 8) | (1 << 12) | (sl << 16) | (3 << 30)
    for w in ws[:8]:
        check(L.sealnn_hgemm_nt(torch.cuda.current_stream(dev).cuda_stream, a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, N, cfg))
    torch.cuda.synchronize()
    bn = {6: 64, 7: 96}[tile]
    wgs = ((M + 319) // 320) * ((N + bn - 1) // bn) * sl
    t = out.flatten()[: wgs * 2 * 192].view(wgs, 2, 192).cpu()
    nk = int(t[0, 0, 182])
    steps = t[:, :, : nk * 3].view(wgs, 2, nk, 3)
    end = t[:, :, 180]
    print(f"tile {tile} M={M} {name} x{sl}: {wgs} workgroups, {nk} steps; shader cycles (s_memtime), mean over workgroups, wave 0 | wave 7")
    print(f"  kernel body ends at {end[:, 0].mean():.0f} | {end[:, 1].mean():.0f} (min {end.min():.0f} max {end.max():.0f})")
    for k in range(nk):
        s = steps[:, :, k]
        nxt = steps[:, :, k + 1, 0] if k + 1 < nk else end
        print(f"  step {k:2d}: top {s[:, 0, 0].mean():7.0f}  vmcnt wait {(s[:, 0, 1] - s[:, 0, 0]).mean():6.0f} | {(s[:, 1, 1] - s[:, 1, 0]).mean():6.0f}   barrier {(s[:, 0, 2] - s[:, 0, 1]).mean():6.0f} | {(s[:, 1, 2] - s[:, 1, 1]).mean():6.0f}"
              f"   reads + MFMAs + issue {(nxt[:, 0] - s[:, 0, 2]).mean():6.0f} | {(nxt[:, 1] - s[:, 1, 2]).mean():6.0f}")
    del ws
