#!/bin/bash
# parity tests + the bench line: tools/quick_round.sh <tag>
tag=$1
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_gputest.log 2>&1
echo "pytest rc=$?"; tail -6 $out/${tag}_gputest.log
timeout 1200 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.log
echo "bench rc=$?"; tail -4 $out/${tag}_bench.log; python - <<'PY' $out/${tag}_bench.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us")}, "mismatches", d["parity_check"]["mismatches"], d["extra"]["phase_ms_one_batch"], "p50", d["extra"]["p50_batch_latency_ms_unpipelined"])
PY
