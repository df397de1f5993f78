#!/bin/bash
# GPU tests, NQ bench line, stress bench line: tools/r3_round.sh <tag> [stress-symbols]
tag=$1
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_gputest.log 2>&1
echo "pytest rc=$?"; tail -8 $out/${tag}_gputest.log
timeout 1200 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.log
echo "bench rc=$?"; tail -3 $out/${tag}_bench.log
if [ -n "$2" ]; then
timeout 1500 python bench.py --workload stress --stress-symbols $2 --steps 5 --warmup 2 > $out/${tag}_stress.json 2> $out/${tag}_stress.log
echo "stress rc=$?"; tail -4 $out/${tag}_stress.log | cut -c1-300
fi
python - <<'PY' $out/${tag}_bench.json $out/${tag}_stress.json
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    pc = d.get("parity_check") or {}
    print(f, {k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "launches", "algorithmic_bytes_per_launch")},
          "mismatches", pc.get("mismatches"), d["extra"].get("phase_ms_one_batch"), (d.get("cpu_baseline") or {}).get("value"))
PY
