#!/bin/bash
# round 4, call 10: the per-token prefix tables (k_constrain_table) -- parity suite, the wide call in isolation with the tables off / on, the line
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/r4_gputest_tables.log 2>&1; echo "pytest rc=$?"; tail -3 $out/r4_gputest_tables.log
for rows in 600 300; do
  EXPAND_NO_COUNT=1 timeout 300 python tools/expand_bench.py --docs 21015324 --rows $rows --prefix-len 1 --iters 30 --variants "SEALFM_PREFIX_TABLES=0|SEALFM_PREFIX_TABLES=1" > $out/r4_tables_ab_${rows}.txt 2>&1; echo "ab $rows rc=$?"
  [ $rows = 600 ] && timeout 300 python tools/expand_bench.py --docs 21015324 --rows $rows --prefix-len 1 --iters 30 --variants "SEALFM_PREFIX_TABLES=0|SEALFM_PREFIX_TABLES=1" > $out/r4_tables_ab_${rows}_counted.txt 2>&1
done
grep -h "variant\|us_per_call" $out/r4_tables_ab_*.txt | cut -c1-400
timeout -s ABRT 600 python -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $out/r4_bench_tables.json 2> $out/r4_bench_tables.log; echo "bench rc=$?"
python - <<'PY' $out/r4_bench_tables.json
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc = d.get("parity_check") or {}
    print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"].get(k) for k in ("frac", "avg_launch_us", "traffic", "algorithmic_bytes_per_launch")}, "mismatches", pc.get("mismatches"), pc.get("values_compared"))
    print("   ", d["extra"].get("phase_ms_one_batch"))
except Exception as e:
    print("no line:", e)
PY
grep "\[bench\]" $out/r4_bench_tables.log | cut -c1-220 | tail -8
