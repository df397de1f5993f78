#!/bin/bash
# GPU tests, then the bench line with the beams' first step shared / not shared: tools/r3_first_ab.sh <tag>
tag=$1
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_gputest.log 2>&1
echo "pytest rc=$?"; tail -4 $out/${tag}_gputest.log
for mode in 1 0; do
  SEAL_SHARED_FIRST_STEP=$mode timeout 900 python bench.py --steps 30 --warmup 3 > $out/${tag}_bench_first$mode.json 2> $out/${tag}_bench_first$mode.log
  echo "bench(shared first step $mode) rc=$?"
  python - <<'PY' $out/${tag}_bench_first$mode.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pc = d.get("parity_check") or {}
print({k: d[k] for k in ("value", "ms_per_step")}, "mismatches", pc.get("mismatches"), d["extra"].get("phase_ms_one_batch"), "p50", d["extra"]["p50_batch_latency_ms_unpipelined"])
PY
  grep "score parity" $out/${tag}_bench_first$mode.log | cut -c1-250
done
