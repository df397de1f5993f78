#!/bin/bash
# round 4, call 17: split-GEMM epilogues applied by the consuming kernels (sealnn_*_acc): GPU tests, the driver's line
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/r4_gputest_acc.log 2>&1; echo "gpu tests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed\|Error" $out/r4_gputest_acc.log | tail -12 | cut -c1-300
timeout -s ABRT 600 python -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $out/r4_bench_acc.json 2> $out/r4_bench_acc.log; echo "bench rc=$?"
python - <<'PY' $out/r4_bench_acc.json
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
tail -3 $out/r4_bench_acc.log | cut -c1-300
