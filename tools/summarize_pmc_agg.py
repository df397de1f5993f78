#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE of the evidence-aggregation kernels (fmi_aggregate.hip + the rocPRIM sorts between them) from two rocprofv3
counter_collection CSVs of bench.py (one --pmc pass each), per batch = per k_agg_locate launch.  JSON on stdout; bench.py cites it
(`roofline_aggregate.traffic`) only while fmi_aggregate.hip is the source it was taken over.
usage: python tools/summarize_pmc_agg.py <fetch.csv> <write.csv> <workload_tag>"""
import csv, hashlib, json, os, sys
from collections import defaultdict

KERNELS = ("k_agg_locate", "k_occ_prepare", "k_mis_prepare", "k_mis", "k_doc_keys", "k_heads", "k_entry_starts", "k_entries", "k_sel_minmax", "k_sel_hist", "k_sel_compact", "k_sel_final",
           "k_select_top", "k_pad_entries", "k_gather", "k_top_docs",
           "k_scatter", "k_full_score", "k_rank_docs", "rocprim")


def reduce(path, counter):
    """per kernel family [launches, sum] -- of the dispatches from the first k_agg_locate on: the index BUILD in front of it launches kernels
    of the same names (k_heads, k_scatter_*, rocPRIM sorts over 2.9 G suffixes) that are not the aggregation's"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
    if key:
        rows.sort(key=lambda r: int(r[key]))
    first = next((i for i, r in enumerate(rows) if "k_agg_locate" in r["Kernel_Name"]), 0)
    acc = defaultdict(lambda: [0, 0.0])
    for r in rows[first:]:
        name = r["Kernel_Name"]
        fam = next((k for k in KERNELS if k in name), None)
        if fam is None:
            continue
        acc[fam][0] += 1
        acc[fam][1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write = reduce(sys.argv[1], "FETCH_SIZE"), reduce(sys.argv[2], "WRITE_SIZE")
    batches = max(1, fetch["k_agg_locate"][0])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {"_aggregate_source_sha256": hashlib.sha256(open(os.path.join(root, "seal_amd", "csrc", "fmi_aggregate.hip"), "rb").read()).hexdigest(),
           "_workload": sys.argv[3] if len(sys.argv) > 3 else None, "batches": batches,
           "_units": "rocprofv3 FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X guide): the *_x2 "
                     "figures double it (right for wide coalesced reads; for 4-byte random gathers the uncorrected figure may already be the bytes that moved)",
           "per_kernel_per_batch": {}}
    tot_f = tot_w = 0.0
    for k in KERNELS:
        f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
        if not f[0] and not w[0]:
            continue
        fm, wm = f[1] * 1024 / batches / 1e6, w[1] * 1024 / batches / 1e6
        tot_f += fm; tot_w += wm
        out["per_kernel_per_batch"][k] = {"launches": f[0] / batches, "fetch_MB": round(fm, 2), "fetch_MB_x2": round(2 * fm, 2), "write_MB": round(wm, 2)}
    out["per_batch_MB"] = {"fetch": round(tot_f, 1), "fetch_x2": round(2 * tot_f, 1), "write": round(tot_w, 1)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
