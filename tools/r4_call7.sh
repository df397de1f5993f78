#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/r4c7_gputest.log 2>&1; echo "gpu tests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed" $out/r4c7_gputest.log | tail -12 | cut -c1-250
python tools/soak_ctl.py final > $out/r4c7_final.txt 2>&1; grep "^==\|CLEAN\|STALL\|differs" $out/r4c7_final.txt | cut -c1-230; grep "\[overlap\]" $out/soak_timing.log | tail -12 | cut -c1-220
timeout -s ABRT 600 python -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $out/r4c7_bench.json 2> $out/r4c7_bench.log; echo "bench rc=$?"
grep "\[bench\]" $out/r4c7_bench.log | cut -c1-250 | tail -12
python - <<'PY' $out/r4c7_bench.json
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc = d.get("parity_check") or {}
    print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "launches", "traffic")}, "mismatches", pc.get("mismatches"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    print(d["extra"].get("phase_ms_one_batch"), d["extra"].get("p50_batch_latency_ms_unpipelined"))
except Exception as e:
    print("no bench line:", e)
PY
