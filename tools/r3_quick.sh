#!/bin/bash
# GPU tests + NQ bench line (+ optional expand_bench): tools/r3_quick.sh <tag> [expand]
tag=$1
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_gputest.log 2>&1
echo "pytest rc=$?"; tail -4 $out/${tag}_gputest.log
timeout 1200 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.log
echo "bench rc=$?"; tail -2 $out/${tag}_bench.log | cut -c1-300
if [ -n "$2" ]; then
  EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --prefix-len 1,2,3 --iters 20 > $out/${tag}_expand_nocount.txt 2> $out/${tag}_expand.err
  EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --prefix-len 3,4,6 --iters 20 --incremental >> $out/${tag}_expand_nocount.txt 2>> $out/${tag}_expand.err
  EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --rows 600 --prefix-len 1,4,6 --iters 20 --incremental >> $out/${tag}_expand_nocount.txt 2>> $out/${tag}_expand.err
  python - <<'PY' $out/${tag}_expand_nocount.txt
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    if "us_per_call" in d: print("   rows", d["rows"], "len", d["prefix_len"], d["us_per_call"], "us", d["avg_allowed_tokens_first8rows"])
PY
fi
python - <<'PY' $out/${tag}_bench.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pc = d.get("parity_check") or {}
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "launches", "algorithmic_bytes_per_launch")},
      "mismatches", pc.get("mismatches"), d["extra"].get("phase_ms_one_batch"), "p50", d["extra"]["p50_batch_latency_ms_unpipelined"])
PY
