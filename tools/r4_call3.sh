#!/bin/bash
# round 4, GPU call 3: GPU tests (all forms of the constraint call, split GEMM, soak), model-side A/B, score parity with the split GEMM, constraint-call A/B
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q > $out/r4c3_gputest.log 2>&1; echo "gpu tests rc=$?"; tail -5 $out/r4c3_gputest.log
python tools/soak_ctl.py ab > $out/r4c3_ab.txt 2>&1; grep "^==\|CLEAN\|STALL\|differs" $out/r4c3_ab.txt | cut -c1-220
SEAL_BENCH_SCORE_PARITY=1 timeout -s ABRT 300 python -X faulthandler bench.py --docs 2000000 --corpus-phrases 2000000 --no-cpu-baseline --steps 10 --warmup 3 > $out/r4c3_bench_quick.json 2> $out/r4c3_bench_quick.log
echo "quick bench rc=$?"; grep "score parity" $out/r4c3_bench_quick.log | cut -c1-300
python - <<'PY' $out/r4c3_bench_quick.json
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "launches")}, d["extra"].get("phase_ms_one_batch"))
except Exception as e:
    print("no bench line:", e)
PY
for rows in 300 600; do
  EXPAND_NO_COUNT=1 timeout 300 python tools/expand_bench.py --rows $rows --prefix-len 2,3,4,6,8 --iters 20 --incremental \
    --variants "SEALFM_SMALL_ROW_MAX=0|SEALFM_SMALL_ROW_MAX=64|SEALFM_SMALL_ROW_MAX=64 SEALFM_ROWS_ONLY_FROM=1|SEALFM_SMALL_ROW_MAX=64 SEALFM_ROW_FIRST=1 SEALFM_ROWS_ONLY_FROM=0" \
    > $out/r4c3_expand_$rows.txt 2> $out/r4c3_expand_$rows.err
  python - <<'PY' $out/r4c3_expand_$rows.txt
import json, sys
v = ""
for line in open(sys.argv[1]):
    try: d = json.loads(line)
    except Exception: continue
    if "variant" in d: v = d["variant"].replace("SEALFM_", ""); continue
    if "us_per_call" in d: print("  rows", d["rows"], "len", d["prefix_len"], "%7.2f us" % d["us_per_call"], " ", v)
PY
done
