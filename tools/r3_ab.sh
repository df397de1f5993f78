#!/bin/bash
# GPU tests, the bench line, then the same bench with the decodes as two loops (no CPU leg): tools/r3_ab.sh <tag>
tag=$1
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_gputest.log 2>&1
echo "pytest rc=$?"; tail -8 $out/${tag}_gputest.log
timeout 1200 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.log
echo "bench rc=$?"; tail -4 $out/${tag}_bench.log
timeout 600 python bench.py --no-joint-decode --no-cpu-baseline > $out/${tag}_bench_nojoint.json 2> $out/${tag}_bench_nojoint.log
echo "bench(no joint) rc=$?"; tail -2 $out/${tag}_bench_nojoint.log
python - <<'PY' $out/${tag}_bench.json $out/${tag}_bench_nojoint.json
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    pc = d.get("parity_check") or {}
    print(f, {k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "launches", "algorithmic_bytes_per_launch")},
          "mismatches", pc.get("mismatches"), d["extra"]["phase_ms_one_batch"], "p50", d["extra"]["p50_batch_latency_ms_unpipelined"])
PY
