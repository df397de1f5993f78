#!/bin/bash
# final check of the tree: GPU tests + the default bench line, both under short limits: tools/r3_final.sh <tag>
tag=$1
out=gpurun_out
mkdir -p $out
timeout -s ABRT 300 python -X faulthandler -m pytest tests -m gpu -x -q > $out/${tag}_gputest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/${tag}_gputest.log | cut -c1-200
timeout -s ABRT 240 python -X faulthandler bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.log
echo "bench rc=$?"
python - <<'PY' $out/${tag}_bench.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pc = d.get("parity_check") or {}
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "launches", "traffic")}, "mismatches", pc.get("mismatches"),
      d["extra"].get("phase_ms_one_batch"), d["extra"].get("decode_step_gemm_algorithms"), "cpu", d["cpu_baseline"]["value"])
PY
grep "score parity" $out/${tag}_bench.log | cut -c1-250
