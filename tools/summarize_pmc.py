#!/usr/bin/env python
"""Reduces a rocprofv3 counter_collection CSV to per-kernel averages (JSON on stdout)."""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: {"launches": len(v), "sum": sum(v), "avg": sum(v) / len(v)} for c, v in cs.items()}
# what the counters were taken over: bench.py only uses a traffic figure whose kernel source is the one it is running
import hashlib
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out["_kernel_source_sha256"] = hashlib.sha256(b"".join(open(os.path.join(root, "seal_amd", "csrc", f), "rb").read()
                                                       for f in ("fmi_kernels.hip", "fmi_device.h", "fmi_internal.h"))).hexdigest()
if len(sys.argv) > 2:
    out["_workload"] = sys.argv[2]          # bench.py's workload_tag: a line only cites counters of its own workload
json.dump(out, sys.stdout, indent=1)
