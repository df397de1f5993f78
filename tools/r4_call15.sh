#!/bin/bash
# round 4, call 15: soak of the final code (100 repetitions of the timed call, every top-100 bit-equal), then the driver's line once more (table sizes in extra)
out=gpurun_out; mkdir -p $out
python tools/soak_ctl.py soak > $out/r4_soak_final.txt 2>&1; grep "^==\|CLEAN\|STALL\|differs\|median" $out/r4_soak_final.txt | cut -c1-230 | tail -8
timeout -s ABRT 600 python -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $out/r4_bench_final.json 2> $out/r4_bench_final.log; echo "bench rc=$?"
python - <<'PY' $out/r4_bench_final.json
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc = d.get("parity_check") or {}
    print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "launches", "traffic", "algorithmic_bytes_per_launch")}, "mismatches", pc.get("mismatches"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    print(d["roofline"].get("traffic_source"))
    print(d["extra"].get("phase_ms_one_batch"), d["extra"].get("p50_batch_latency_ms_unpipelined"), d["extra"].get("prefix_tables"))
except Exception as e:
    print("no bench line:", e)
PY
