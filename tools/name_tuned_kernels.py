"""Which hipBLASLt kernels do the shipped TunableOp picks (tools/tuned_bisect/all_r3_shipped.csv) run?  Run under
`rocprofv3 --kernel-trace --stats`: every shape of the file is issued once through F.linear with TunableOp reading the file, then once
more with TunableOp off (the library's own heuristic pick); the kernel names are in the trace, in this order."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
import torch.cuda.tunable as tn
dev = torch.device("cuda:0")
shapes = [(600, 3072, 1024), (600, 4096, 1024), (600, 1024, 4096), (600, 50265, 1024), (300, 4096, 1024), (300, 1024, 4096), (300, 50265, 1024)]
ws = {}
for M, N, K in shapes:
    ws[(M, N, K)] = (torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev))
torch.cuda.synchronize()
for mode in ("tuned", "default"):
    if mode == "tuned":
        tn.enable(True); tn.tuning_enable(False); tn.record_untuned_enable(False)
        print("read_file:", tn.read_file(os.path.join(os.path.dirname(__file__), "tuned_bisect", "all_r3_shipped.csv")))
    else:
        tn.enable(False)
    for (M, N, K), (x, w, b) in ws.items():
        marker = torch.zeros(M + (7 if mode == "tuned" else 13), device=dev)     # a fill kernel of a telling size between the GEMMs
        y = F.linear(x, w, b)
        torch.cuda.synchronize()
        print(mode, (M, N, K), float(y.abs().mean()))
