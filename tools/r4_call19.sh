#!/bin/bash
# round 4, call 19: cross-attention kernels (batched staging, branch-free loops, the rescoring forward through runs of 16 rows): GPU tests, the driver's line
out=gpurun_out; mkdir -p $out
timeout 300 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/r4_gputest_ca.log 2>&1; echo "gpu tests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed\|Error" $out/r4_gputest_ca.log | tail -8 | cut -c1-300
timeout -s ABRT 400 python -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $out/r4_bench_ca.json 2> $out/r4_bench_ca.log; echo "bench rc=$?"
python - <<'PY' $out/r4_bench_ca.json
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc = d.get("parity_check") or {}
    print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "traffic")}, "mismatches", pc.get("mismatches"), pc.get("values_compared"))
    print(d["extra"].get("phase_ms_one_batch"), d["extra"].get("p50_batch_latency_ms_unpipelined"))
    bs = pc["by_kind"]; print(bs["beam_scores"]["max_abs_err"], bs["rescore_scores"]["max_abs_err"])
except Exception as e:
    print("no bench line:", e)
PY
