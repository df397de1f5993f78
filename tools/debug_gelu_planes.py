"""Which elements does sealnn_gelu_planes split differently from split_planes_reference(torch gelu)?  (tests/test_split_gemm.py)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from seal_amd._lib import check, lib
from seal_amd.split_gemm import LO_SHIFT, split_planes_reference
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator().manual_seed(9)
rows = 77
torch.randn(rows, 1024, generator=g); torch.randn(rows, 1024, generator=g); torch.rand(1024, generator=g); torch.randn(1024, generator=g)
h = (torch.randn(rows, 4096, generator=g) * 2).to(dev)
hp = torch.empty(rows, 3 * 4096, dtype=torch.float16, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
check(lib().sealnn_gelu_planes(st, h.data_ptr(), rows, 4096, hp.data_ptr(), flag.data_ptr()))
gt = torch.nn.functional.gelu(h)
ge = 0.5 * h * (1 + torch.erf(h * 0.70710678118654752440))
ref = split_planes_reference(gt.cpu())
got = hp.cpu()
back = got[:, :4096].double() + got[:, 8192:].double() * 2.0 ** -LO_SHIFT
want = ref[:, :4096].double() + ref[:, 8192:].double() * 2.0 ** -LO_SHIFT
d = (back - want).abs()
print("max diff", d.max().item(), "elements over 1e-6:", int((d > 1e-6).sum()), "of", d.numel(), "| torch gelu vs erf formula max", (gt - ge).abs().max().item())
idx = torch.nonzero(d > 1e-6)[:12]
for r, c in idx.tolist():
    print("x=%r gelu_torch=%r | gpu hi=%r lo=%r | ref hi=%r lo=%r | back-gelu=%.3e" % (h[r, c].item(), gt[r, c].item(), got[r, c].item(), got[r, 8192 + c].item(),
          ref[r, c].item(), ref[r, 8192 + c].item(), back[r, c].item() - gt[r, c].double().item()))
