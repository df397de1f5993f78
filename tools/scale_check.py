#!/usr/bin/env python
"""Size-independent property check of a GPU-built index at full scale (no oracle can follow there):
for patterns cut out of random documents, every located row must (a) bin to a document whose text
really contains the pattern at that position and (b) the pattern's own document must be among them;
counts must equal hi-lo; get_doc must return the generated tokens.

  python tools/scale_check.py --docs 21015324      # NQ size
  python tools/scale_check.py --docs 36000000      # KILT size (> 2^32 symbols: 64-bit builder path)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=2000000)
    ap.add_argument("--patterns", type=int, default=300)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from seal_amd import FMIndex
    t0 = time.perf_counter()
    data, beg, title_len, _ = bench.synth_corpus(args.docs, dev, seed=0)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    n_docs = beg.numel() - 1
    picks = []
    for _ in range(args.patterns):
        d = int(rng.integers(0, n_docs))
        b, e = int(beg[d]), int(beg[d + 1])
        fwd = torch.flip(data[b:e].long() - bench.SHIFT, [0]).tolist()
        m = int(rng.integers(1, 6))
        a = int(rng.integers(0, len(fwd) - m))
        picks.append((d, fwd, a, fwd[a:a + m]))
    t0 = time.perf_counter()
    ix = FMIndex()
    ix.initialize_from_device(data, beg.tolist())
    del data
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    n = ix.size()
    lo, hi = ix.get_range_batch([p[3] for p in picks])
    bad = 0
    checked_rows = 0
    for (d, fwd, a, pat), l, h in zip(picks, lo.tolist(), hi.tolist()):
        cnt = h - l
        assert cnt >= 1, ("pattern from the corpus not found", pat)
        rows = np.arange(l, min(h, l + 64), dtype=np.uint64)
        pos, docs = ix.locate_batch(rows)
        found_own = cnt > 64
        for p, dd in zip(pos.tolist(), docs.tolist()):
            checked_rows += 1
            # position p is the start of reversed(pattern) in the reversed text of document dd
            tb, te = ix.beginnings[dd], ix.beginnings[dd + 1]
            off = te - 1 - p                      # index of the pattern's LAST token in forward order
            doc_tokens = ix.get_doc(dd)
            s = off - (len(pat) - 1)
            if doc_tokens[s:s + len(pat)] != pat:
                # a Q1-widened single-token range may carry one foreign row
                if not (len(pat) == 1 and cnt > 1):
                    bad += 1
            if dd == d:
                found_own = True
        assert found_own, ("own document not among the located rows", d, pat)
    # ---- the constraint kernel (k_constrain) at this scale: for decoder prefixes cut out of the corpus, a token is allowed
    # iff the prefix extended by it occurs -- checked for EVERY token of the vocabulary on a few rows (50 k counts each,
    # one launch) and for the allowed tokens plus the corpus's own continuation on all rows
    import ctypes
    from seal_amd._lib import check, lib
    V = bench.VOCAB
    mask_rows = []
    for d, fwd, a, pat in picks[:96]:
        if a + len(pat) < len(fwd):
            mask_rows.append((pat, fwd[a + len(pat)]))
    width = max(len(p) for p, _ in mask_rows)
    by_len = {}
    for pat, nxt in mask_rows:
        by_len.setdefault(len(pat), []).append((pat, nxt))
    mask_checked = mask_bad = full_rows = 0
    for plen, group in sorted(by_len.items()):
        ids = torch.tensor([[2] + pat for pat, _ in group], dtype=torch.long, device=dev)
        bits = torch.zeros(len(group), (V + 31) // 32, dtype=torch.int32, device=dev)
        check(lib().fmi_dev_allowed_bits(ix.handle, torch.cuda.current_stream(dev).cuda_stream, len(group), plen + 1, ids.data_ptr(),
                                         bits.data_ptr(), V, bench.SHIFT, 1, 2, None, 0, 0, 0))
        torch.cuda.synchronize()
        words = bits.cpu().numpy().view(np.uint32)
        allowed = np.unpackbits(words.view(np.uint8), axis=1, bitorder="little")[:, :V].astype(bool)
        for r, (pat, nxt) in enumerate(group):
            assert allowed[r, nxt], ("the corpus's own continuation is not allowed", pat, nxt)
            toks = np.nonzero(allowed[r])[0].tolist()
            if full_rows < 4 and plen <= 2:                    # the whole vocabulary, both directions
                cand = list(range(V))
                full_rows += 1
            else:
                cand = toks[:200]
            l2, h2 = ix.get_range_batch([pat + [t] for t in cand])
            occurs = (h2.astype(np.int64) - l2.astype(np.int64)) > 0
            for t, oc in zip(cand, occurs.tolist()):
                mask_checked += 1
                if bool(allowed[r, t]) != bool(oc) and t not in (0, 1, 2, 3):
                    mask_bad += 1
    assert full_rows >= 1 and mask_bad == 0, (full_rows, mask_bad)
    sample_docs = [0, 1, n_docs // 2, n_docs - 1]
    for d, fwd, _, _ in picks[:20]:
        assert ix.get_doc(d) == fwd
    print(json.dumps({"docs": args.docs, "n": n, "wide_indices": n > 2**32, "hbm_gib": round(ix.device_bytes() / 2**30, 2),
                      "corpus_s": round(t_gen, 1), "build_s": round(t_build, 1), "patterns": len(picks), "rows_checked": checked_rows,
                      "mismatching_rows": bad, "mask_rows": len(mask_rows), "mask_tokens_checked": mask_checked, "mask_rows_over_the_whole_vocabulary": full_rows,
                      "mask_mismatches": mask_bad, "peak_torch_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
    assert bad == 0


if __name__ == "__main__":
    main()
