import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from seal_amd import FMIndex
from seal_amd.beam_search import IndexBasedLogitsProcessor
dev = torch.device("cuda:0")
data, beg, tl, ids_by_rank = bench.synth_corpus(int(os.environ.get("DOCS", 2000000)), dev)
N = data.numel()
g = torch.Generator(device=dev); g.manual_seed(1)
index = FMIndex(); 
B, K, V = 20, 15, bench.VOCAB
ids_all = {}
for cl in (1, 2, 3, 6):
    p = torch.randint(cl + 1, N - 1, (B * K,), generator=g, device=dev)
    offs = torch.arange(cl - 1, device=dev)
    toks = data[(p[:, None] - offs[None, :])].long() - bench.SHIFT
    ids_all[cl] = torch.cat([torch.full((B * K, 1), 2, device=dev, dtype=torch.long), toks], 1).contiguous()
index.initialize_from_device(data, beg.tolist()); del data
proc = IndexBasedLogitsProcessor(index, K, pad_token_id=1, eos_token_id=2)
logits = torch.randn(B * K, V, device=dev)
bs = torch.zeros(B * K, device=dev)
for cl in (1, 2, 3, 6):
    ids = ids_all[cl]
    for _ in range(2): proc.fused_topk(ids, logits, bs, B, K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): proc.fused_topk(ids, logits, bs, B, K)
    torch.cuda.synchronize(); print("cur_len", cl, "fused ms", (time.perf_counter() - t0) * 100, file=sys.stderr)
