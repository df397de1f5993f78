#!/usr/bin/env python
"""Isolated measurement of the rank/select constraint kernel (k_constrain: prefix
range + expansion, one launch) on an NQ-shaped synthetic index, model-free ("trace mode"):
rows are decoder prefixes drawn from the corpus itself.

  python tools/expand_bench.py --docs 21015324 --rows 300 --prefix-len 1 --iters 20

prints one JSON line: probes, algorithmic bytes (128 B per block probed), HIP-event time,
GB/s and fraction of the 8 TB/s HBM peak.  Run it under
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/expand_bench.py ...
for the memory-side traffic of the same launches."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=21015324)
    ap.add_argument("--rows", type=int, default=300)
    ap.add_argument("--prefix-len", type=str, default="1", help="tokens after the decoder start token; comma-separated list = "
                    "one measurement per length on the same index")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--synthetic-bwt", type=float, default=0,
                    help="N symbols: skip corpus+suffix array and load an i.i.d. Zipf 'BWT' of N symbols straight into the "
                         "wavelet matrix (rank/select-only index; bandwidth measurement only, SURVEY.md 8d tier X)")
    args = ap.parse_args()
    plens = [int(x) for x in args.prefix_len.split(",")]
    dev = torch.device("cuda", 0)
    from seal_amd import FMIndex
    from seal_amd._lib import check, lib
    g = torch.Generator(device=dev)
    g.manual_seed(100 + args.seed)
    if args.synthetic_bwt:
        N = int(args.synthetic_bwt)
        usable = torch.arange(4, bench.VOCAB, device=dev)
        ids_by_rank = usable[torch.randperm(usable.numel(), generator=g, device=dev)]
        w = 1.0 / torch.arange(1, usable.numel() + 1, device=dev, dtype=torch.float64) ** 1.07
        cdf = torch.cumsum(w, 0) / w.sum()
        bwt = torch.empty(N, dtype=torch.int16, device=dev)
        for a in range(0, N, 1 << 27):
            b = min(N, a + (1 << 27))
            r = torch.searchsorted(cdf, torch.rand(b - a, generator=g, device=dev, dtype=torch.float64)).clamp_(max=usable.numel() - 1)
            bwt[a:b] = (ids_by_rank[r] + bench.SHIFT).to(torch.int16)      # two's complement view of the u16 symbol
        bwt[N // 3] = 0
        all_ids = []
        for pl in plens:
            r = torch.searchsorted(cdf, torch.rand(args.rows * pl, generator=g, device=dev, dtype=torch.float64))
            toks = ids_by_rank[r.clamp_(max=usable.numel() - 1)].view(args.rows, pl)
            all_ids.append(torch.cat([torch.full((args.rows, 1), 2, device=dev, dtype=torch.long), toks], 1).contiguous())
        index = FMIndex()
        index.initialize_rank_only_from_bwt(bwt, bench.VOCAB - 1 + bench.SHIFT)
        del bwt
        data = None
    else:
        data, beg, title_len, ids_by_rank = bench.synth_corpus(args.docs, dev, seed=0)
    if data is not None:
        # prefixes = corpus n-grams in forward order: pick positions in the reversed text and read backwards
        N = data.numel()
        all_ids = []
        for pl in plens:
            p = torch.randint(pl + 1, N - 1, (args.rows,), generator=g, device=dev)
            offs = torch.arange(pl, device=dev)
            toks = data[(p[:, None] - offs[None, :])].long() - bench.SHIFT          # forward order
            all_ids.append(torch.cat([torch.full((args.rows, 1), 2, device=dev, dtype=torch.long), toks], 1).contiguous())
        index = FMIndex()
        index.initialize_from_device(data, beg.tolist())
        del data
    h = index.handle
    V = bench.VOCAB
    bits = torch.zeros(args.rows, (V + 31) // 32, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    check(lib().fmi_dev_enable_probe_count(h, 0 if os.environ.get("EXPAND_NO_COUNT") else 1))
    check(lib().fmi_dev_enable_timing(h, 1))

    for pl, ids in zip(plens, all_ids):
        def call():
            check(lib().fmi_dev_allowed_bits(h, st, args.rows, ids.shape[1], ids.data_ptr(), bits.data_ptr(), V, bench.SHIFT, 1, 2,
                                             None, 0, 0, 0))
        call()
        torch.cuda.synchronize()
        probes, launches, ms = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double()
        if not os.environ.get("EXPAND_NO_COUNT"):
            check(lib().fmi_dev_read_probe_count(h, ctypes.byref(probes)))
        check(lib().fmi_dev_read_timing(h, ctypes.byref(launches), ctypes.byref(ms)))
        for _ in range(args.iters):
            call()
        stats = (ctypes.c_uint64 * 4)()
        if not os.environ.get("EXPAND_NO_COUNT"):
            check(lib().fmi_dev_read_expand_stats(h, stats))
            check(lib().fmi_dev_read_probe_count(h, ctypes.byref(probes)))
        check(lib().fmi_dev_read_timing(h, ctypes.byref(launches), ctypes.byref(ms)))
        allowed = int(sum(bin(x & 0xFFFFFFFF).count("1") for x in bits[:8].flatten().tolist())) / 8.0
        gbs = probes.value * 128 / (ms.value * 1e-3) / 1e9
        print(json.dumps({"docs": args.docs, "n": index.size(), "rows": args.rows, "prefix_len": pl,
                          "iters": args.iters, "blocks_per_call": probes.value / args.iters,
                          "alg_MB_per_call": round(probes.value * 128 / args.iters / 1e6, 2),
                          "us_per_call": round(ms.value * 1e3 / args.iters, 2), "GBps": round(gbs, 1),
                          "frac_of_8TBps": round(gbs / 8000, 4), "wave_iters_per_call": stats[1] / args.iters,
                          "lane_pair_util": round(stats[2] / max(1, 32 * stats[1]), 3), "model_probes_per_call": 2 * stats[3] / args.iters,
                          "model_GBps": round(2 * stats[3] * 64 / (ms.value * 1e-3) / 1e9, 1), "avg_allowed_tokens_first8rows": allowed}), flush=True)


if __name__ == "__main__":
    main()
