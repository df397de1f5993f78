#!/bin/bash
# round 4, call 13: final evidence with the prefix tables in: GPU tests, rocprofv3 kernel stats + PMC pass of bench.py, per-call numbers, the driver's line
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/r4_gputest_final.log 2>&1; echo "gpu tests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed" $out/r4_gputest_final.log | tail -12 | cut -c1-250
bash tools/prof_bench.sh $out r4 > $out/r4_prof_bench.log 2>&1; echo "prof rc=$?"; ls $out | grep "^r4_" | head -30
timeout 300 python tools/prof_constrain_steps.py > $out/r4_constrain_per_call.txt 2> $out/r4_constrain_per_call.err; echo "per-call rc=$?"; tail -20 $out/r4_constrain_per_call.txt | cut -c1-200
timeout -s ABRT 600 python -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $out/r4_bench_final.json 2> $out/r4_bench_final.log; echo "bench rc=$?"
grep "\[bench\]" $out/r4_bench_final.log | cut -c1-250 | tail -8
python - <<'PY' $out/r4_bench_final.json
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc = d.get("parity_check") or {}
    print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "launches", "traffic", "algorithmic_bytes_per_launch")}, "mismatches", pc.get("mismatches"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    print(d["roofline"].get("traffic_source"))
    print(d["extra"].get("phase_ms_one_batch"), d["extra"].get("p50_batch_latency_ms_unpipelined"))
except Exception as e:
    print("no bench line:", e)
PY
