#!/usr/bin/env python
"""Where a batch's GPU time goes, from a rocprofv3 kernel trace of bench.py (tools/prof_bench.sh keeps it as gpurun_out/<tag>_kernel_trace_full.csv):
busy time and launches per batch by kernel family over the TIMED batches (from the second k_constrain_table launch -- the first belongs to the
warm-up -- to the first one after them), and the strided fp32 copies torch puts in front of GEMMs (what a copy precedes).
Then the same for a decode's PREFIX alone (encoder, cross-attention K / V, first step): the kernels on the decode queue between the last k_beam_advance of
the first and the last library GEMM a batch has on the decode queue (a decode's later steps hold none), extended to the bookkeeping launches either side.
usage: python tools/trace_by_category.py <kernel_trace.csv> [timed batches, default 3]"""
import collections
import csv
import re
import sys

FAMILIES = ["k_constrain", "k_table_bits", "k_beam_advance", "k_hgemm", "k_row_pick", "k_query_merge", "k_self_attn", "k_tree_self", "k_cross_attn", "k_add_layernorm", "k_gelu",
            "k_split_planes", "k_entries", "k_agg", "k_full_score", "k_mis", "rocprim"]
TORCH = r"(direct_copy|CUDAFunctor_add|index_elementwise|_scatter_gather|vectorized_elementwise_kernel|layer_norm|gather|indexSelect|reduce_kernel|SoftMax|sort|topk|fill|cat|where|masked|arange|cumsum)"


def family(name):
    if "Cijk" in name:
        return "library GEMM, fp16 in / fp32 out" if "HSS" in name else "library GEMM, fp32"
    for k in FAMILIES:
        if k in name:
            return k
    if "at::native" in name:
        m = re.search(TORCH, name)
        return "torch: " + (m.group(1) if m else name[20:70])
    return "other: " + name[:40]


def main():
    path = sys.argv[1]
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]), r["Queue_Id"]))
    rows.sort()
    first = [i for i, r in enumerate(rows) if "k_constrain_table" in r[2]]
    if len(first) < batches + 2:
        sys.exit("the trace holds %d table calls: not a bench.py run of >= %d timed batches" % (len(first), batches))
    t0, t1 = rows[first[1]][0], rows[first[1 + batches]][0]
    win = [r for r in rows if t0 <= r[0] < t1]
    acc = collections.defaultdict(lambda: [0, 0])
    for r in win:
        a = acc[family(r[2])]
        a[0] += r[1] - r[0]
        a[1] += 1
    busy = sum(v[0] for v in acc.values())
    print("%d kernels in %d timed batches: busy %.2f ms per batch, wall %.2f ms per batch (under the tracer)" % (len(win), batches, busy / batches / 1e6, (t1 - t0) / batches / 1e6))
    for k, v in sorted(acc.items(), key=lambda x: -x[1][0])[:30]:
        print("  %-55s %8.3f ms/batch %8.1f launches/batch %7.1f us each" % (k, v[0] / batches / 1e6, v[1] / batches, v[0] / v[1] / 1e3))
    print("strided copies (torch direct_copy, not vectorised) by element type, grid, queue and the kernel that follows on the queue:")
    cp = collections.defaultdict(lambda: [0, 0])
    for i, r in enumerate(win):
        if "direct_copy" in r[2] and "vectorized" not in r[2]:
            m = re.search(r"lambda\((\w+( \w+)?)\)#1\}>", r[2])
            nxt = next((x[2] for x in win[i + 1:i + 30] if x[4] == r[4]), "")
            key = (m.group(1) if m else "?", r[3], r[4], re.sub(r"<.*", "", nxt.replace("void ", ""))[:44])
            cp[key][0] += r[1] - r[0]
            cp[key][1] += 1
    for k, v in sorted(cp.items(), key=lambda x: -x[1][0])[:12]:
        print("  %-90s %7.3f ms/batch %6.1f/batch" % (str(k), v[0] / batches / 1e6, v[1] / batches))
    # the decode prefix of every timed batch but the first
    dq = rows[first[1]][4]
    on_q = [r for r in rows if r[4] == dq]
    tables = [i for i, r in enumerate(on_q) if "k_constrain_table" in r[2]]
    pacc, walls, n = collections.defaultdict(lambda: [0, 0]), [], 0
    for t_prev, ti in zip(tables[1:1 + batches], tables[2:2 + batches]):
        # (a decode's steps hold no library GEMM: the prefix is where the queue's Cijk kernels are, up to the bookkeeping launch behind the first step)
        lib_ix = [i for i in range(t_prev, ti) if "Cijk" in on_q[i][2]]
        if not lib_ix:
            continue
        j0 = lib_ix[0]
        while j0 > t_prev and not any(k in on_q[j0 - 1][2] for k in ("k_beam_advance", "k_row_pick", "k_query_merge", "k_hgemm")):
            j0 -= 1
        j1 = lib_ix[-1]
        while j1 + 1 < ti and "k_beam_advance" not in on_q[j1][2]:
            j1 += 1
        seg = on_q[j0:j1 + 1]
        n += 1
        walls.append((seg[-1][1] - seg[0][0]) / 1e6)
        for r in seg:
            a = pacc[family(r[2])]
            a[0] += r[1] - r[0]
            a[1] += 1
    if n:
        print("decode prefix (encoder, cross K / V, first step) on queue %s: busy %.2f ms, %.0f kernels per batch; first to last kernel %.1f ms (a batch's encoder is "
              "enqueued ahead of its first step: the two are not contiguous on the queue)" %
              (dq, sum(v[0] for v in pacc.values()) / n / 1e6, sum(v[1] for v in pacc.values()) / n, sum(walls) / n))
        for k, v in sorted(pacc.items(), key=lambda x: -x[1][0])[:16]:
            print("  %-55s %8.3f ms/batch %8.1f launches/batch %7.1f us each" % (k, v[0] / n / 1e6, v[1] / n, v[0] / v[1] / 1e3))


if __name__ == "__main__":
    main()
