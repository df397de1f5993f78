#!/bin/bash
# tuned GEMM picks (seal_amd/tuned_gemm.py): live tuning run, then the bench line with the shipped file and without: tools/r3_tune_ab.sh <tag>
tag=$1
out=gpurun_out
mkdir -p $out
rm -f $out/${tag}_tuned_live.csv
t0=$(date +%s)
SEAL_TUNED_GEMMS=tune:$out/${tag}_tuned_live.csv timeout 900 python bench.py --steps 4 --warmup 2 > $out/${tag}_bench_tune.json 2> $out/${tag}_bench_tune.log
echo "tune run rc=$? $(( $(date +%s) - t0 )) s"; cat $out/${tag}_tuned_live.csv | cut -c1-120
for mode in file 0; do
  SEAL_TUNED_GEMMS=$mode timeout 900 python bench.py --steps 30 --warmup 3 > $out/${tag}_bench_$mode.json 2> $out/${tag}_bench_$mode.log
  echo "bench($mode) rc=$?"
  python - <<'PY' $out/${tag}_bench_$mode.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pc = d.get("parity_check") or {}
print({k: d[k] for k in ("value", "ms_per_step")}, "mismatches", pc.get("mismatches"), d["extra"].get("phase_ms_one_batch"), "p50", d["extra"]["p50_batch_latency_ms_unpipelined"])
PY
  grep "score parity" $out/${tag}_bench_$mode.log | cut -c1-250
done
