#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 120 python tools/debug_gelu_planes.py > $out/r4c6_gelu.txt 2>&1; grep "max diff" $out/r4c6_gelu.txt | cut -c1-200
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/r4c6_gputest.log 2>&1; echo "gpu tests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed" $out/r4c6_gputest.log | tail -12 | cut -c1-250
python tools/soak_ctl.py excl3 > $out/r4c6_excl3.txt 2>&1; grep "^==\|CLEAN\|STALL\|differs" $out/r4c6_excl3.txt | cut -c1-230
timeout -s ABRT 840 python -X faulthandler bench.py --workload stress --steps 5 --warmup 2 > $out/r4c6_stress.json 2> $out/r4c6_stress.log; echo "stress rc=$?"
grep "\[bench\]" $out/r4c6_stress.log | cut -c1-260 | tail -25; tail -3 $out/r4c6_stress.log | cut -c1-300
python - <<'PY' $out/r4c6_stress.json
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc = d.get("parity_check") or {}
    print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"].get("frac"), "mismatches", pc.get("mismatches"), {k: (v.get("values"), v.get("mismatches")) for k, v in (pc.get("by_kind") or {}).items()})
    print(d["config"]["workload"][:200], d["config"]["index_hbm_gib"], d["extra"].get("phase_ms_one_batch"), d.get("cpu_baseline", {}) and d["cpu_baseline"].get("value"))
except Exception as e:
    print("no stress line:", e)
PY
