"""one product of a decode step, 40 launches over rotating weights, for a rocprofv3 --pmc pass (tools/pmc_hgemm.sh): python tools/hgemm_pmc_driver.py <shape> [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seal_amd._lib import check, lib
from seal_amd import split_gemm
dev = torch.device("cuda:0")
N, K = {"dxd": (1024, 1024), "qkv": (3072, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096), "lm_head": (50265, 1024)}[sys.argv[1]]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 600
cfg = split_gemm.hand_config(M, N, 3 * K, True) | split_gemm.PAIRS_BIT
a = torch.randn(M, 2 * K, device=dev).half()
ws = [torch.randn(N, 2 * K, device=dev).half() for _ in range(max(2, int(640e6 / (N * K * 4))))]
out = torch.empty(16, M, N, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for i in range(40):
    w = ws[i % len(ws)]
    check(lib().sealnn_hgemm_nt(st, a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, 2 * K, N, cfg))
torch.cuda.synchronize()
print(sys.argv[1], M, hex(cfg))
