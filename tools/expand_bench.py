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
    ap.add_argument("--incremental", action="store_true", help="time the call of a decode step: the prefix of length L-1 was "
                    "searched by the previous call (kept ranges, workspace bitmaps), the timed call extends it by one token")
    ap.add_argument("--timestamps", action="store_true", help="per-wave realtime stamps of one call (where the time goes)")
    ap.add_argument("--variants", default="", help="'|'-separated settings, each a space-separated list of SEALFM_<OPTION>=VALUE (fmi_dev_set_option: the "
                    "launch-shape switches of the constraint call): the measurements are repeated for each on the same index")
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

    ident = torch.arange(args.rows, device=dev)
    tsbuf = torch.zeros(args.rows * 16 * 8, dtype=torch.int64, device=dev) if args.timestamps else None
    for variant in (args.variants.split("|") if args.variants else [""]):
        for kv in variant.split():
            k, v = kv.split("=", 1)
            if k.startswith("SEALFM_"):        # launch-shape switches: per handle since round 4 (the environment is read at handle creation)
                check(lib().fmi_dev_set_option(h, k[len("SEALFM_"):].lower().encode(), int(v)))
            else:
                os.environ[k] = v
        if args.variants:
            print(json.dumps({"variant": variant}), flush=True)
        for pl, ids in zip(plens, all_ids):
            def call():
                if not args.incremental:
                    check(lib().fmi_dev_allowed_bits(h, st, args.rows, ids.shape[1], ids.data_ptr(), bits.data_ptr(), V, bench.SHIFT, 1, 2,
                                                     None, 0, 0, 0))
                    return
                # step t-1 (untimed), then step t as the decode loop issues it
                check(lib().fmi_dev_enable_timing(h, 0))
                if ids.shape[1] > 2:
                    prev = ids[:, :-1].contiguous()
                    check(lib().fmi_dev_allowed_bits_step(h, st, args.rows, prev.shape[1], prev.data_ptr(), None, V, bench.SHIFT, 1, 2, None, 0, 0, 0,
                                                          77, None, None))
                check(lib().fmi_dev_enable_timing(h, 1))
                if not os.environ.get("EXPAND_NO_COUNT"):
                    junk = ctypes.c_uint64()
                    check(lib().fmi_dev_read_probe_count(h, ctypes.byref(junk)))       # drop the untimed step's counts
                out = ctypes.c_void_p()
                check(lib().fmi_dev_allowed_bits_step(h, st, args.rows, ids.shape[1], ids.data_ptr(), None, V, bench.SHIFT, 1, 2, None, 0, 0, 0,
                                                      77, ident.data_ptr() if ids.shape[1] > 2 else None, ctypes.byref(out)))
                torch.cuda.synchronize()
                l1, m1 = ctypes.c_uint64(), ctypes.c_double()
                check(lib().fmi_dev_read_timing(h, ctypes.byref(l1), ctypes.byref(m1)))
                acc[0] += m1.value
                if not os.environ.get("EXPAND_NO_COUNT"):
                    xs = (ctypes.c_uint64 * 4)()
                    check(lib().fmi_dev_read_expand_stats(h, xs))                       # the timed call's own counts
                    for j in range(4):
                        tot[j] += xs[j]
                from bench import _CudaArray
                ws = torch.as_tensor(_CudaArray(out.value, args.rows * ((V + 31) // 32), "<i4"), device=dev)
                bits.copy_(ws.view(args.rows, -1))
            acc = [0.0]
            tot = [0, 0, 0, 0]
            call()
            torch.cuda.synchronize()
            probes, launches, ms = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double()
            if not os.environ.get("EXPAND_NO_COUNT"):
                check(lib().fmi_dev_read_probe_count(h, ctypes.byref(probes)))
            check(lib().fmi_dev_read_timing(h, ctypes.byref(launches), ctypes.byref(ms)))
            acc[0] = 0.0
            tot = [0, 0, 0, 0]
            for _ in range(args.iters):
                call()
            stats = (ctypes.c_uint64 * 4)()
            if not os.environ.get("EXPAND_NO_COUNT"):
                check(lib().fmi_dev_read_expand_stats(h, stats))
                check(lib().fmi_dev_read_probe_count(h, ctypes.byref(probes)))
            check(lib().fmi_dev_read_timing(h, ctypes.byref(launches), ctypes.byref(ms)))
            if args.incremental:
                ms.value = acc[0]
                if not os.environ.get("EXPAND_NO_COUNT"):
                    probes.value = tot[0]
                    for j in range(4):
                        stats[j] = tot[j]
            if args.timestamps:
                check(lib().fmi_dev_debug_timestamps(h, tsbuf.data_ptr(), tsbuf.numel()))
                tsbuf.zero_()
                call()
                torch.cuda.synchronize()
                check(lib().fmi_dev_debug_timestamps(h, None, 0))
                t = tsbuf.view(-1, 8).cpu().numpy().astype("int64")
                if os.environ.get("EXPAND_STAMP_DIR"):
                    __import__("numpy").save(os.path.join(os.environ["EXPAND_STAMP_DIR"], f"stamps_len{pl}.npy"), t)
                t = t[t[:, 0] > 0]
                t0 = t[:, 0].min()
                def q(col, mask=None):
                    v = (t[:, col] - t0)[(t[:, col] > 0) if mask is None else mask] / 100.0        # 100 MHz -> us
                    np_ = __import__("numpy")
                    return None if v.size == 0 else [round(float(x), 2) for x in (v.min(), float(np_.median(v)), float(np_.percentile(v, 90)), v.max())]
                print(json.dumps({"prefix_len": pl, "rows": args.rows, "waves": int(t.shape[0]), "us_since_first_wave_start[min,median,p90,max]": {
                    "wave_start": q(0), "prefix_range_known": q(1), "root_child_known(non-empty only)": q(2), "leaf_phase_start(workgroup)": q(5), "subtree_done": q(3), "bitmap_stored": q(4)},
                    "waves_with_a_subtree": int((t[:, 2] > 0).sum())}), flush=True)
            allowed = int(sum(bin(x & 0xFFFFFFFF).count("1") for x in bits[:8].flatten().tolist())) / 8.0
            gbs = probes.value * 128 / (ms.value * 1e-3) / 1e9
            print(json.dumps({"docs": args.docs, "n": index.size(), "rows": args.rows, "prefix_len": pl,
                              "iters": args.iters, "blocks_per_call": probes.value / args.iters,
                              "alg_MB_per_call": round(probes.value * 128 / args.iters / 1e6, 2),
                              "us_per_call": round(ms.value * 1e3 / args.iters, 2), "GBps": round(gbs, 1),
                              "frac_of_8TBps": round(gbs / 8000, 4), "wave_iters_per_call": stats[1] / args.iters,
                              "lane_pair_util": round(stats[2] / max(1, 32 * stats[1]), 3), "model_probes_per_call": 2 * stats[3] / args.iters,
                              "model_GBps": round(2 * stats[3] * 64 / (ms.value * 1e-3) / 1e9, 1), "avg_allowed_tokens_first8rows": allowed,
                              "bitmap_checksum": int((bits.long() * (1 + torch.arange(bits.numel(), device=dev).view_as(bits) % 8191)).sum())}), flush=True)


if __name__ == "__main__":
    main()
