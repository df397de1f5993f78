#!/bin/bash
# A/B of the row-first constraint call (k_constrain_rows + k_constrain) against the single launch, one box
out=gpurun_out; tag=$1
timeout 600 python -m pytest tests/test_gpu_fmindex.py tests/test_gpu_decode.py tests/test_reference_golden.py tests/test_gpu_bench_parity.py -m gpu -x -q 2>&1 | tail -3
SEALFM_ROW_FIRST=1 timeout 600 python -m pytest tests/test_gpu_fmindex.py tests/test_gpu_decode.py tests/test_reference_golden.py -m gpu -x -q 2>&1 | tail -3
V='SEALFM_ROW_FIRST=0|SEALFM_ROW_FIRST=1|SEALFM_ROW_FIRST=0|SEALFM_ROW_FIRST=1'
for rows in 300 600; do
  EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --rows $rows --prefix-len 2,3,4,6 --iters 20 --incremental --variants "$V" >> $out/${tag}_ab.txt 2>> $out/${tag}_ab.err
done
EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --rows 300 --prefix-len 1 --iters 20 --variants "$V" >> $out/${tag}_ab.txt 2>> $out/${tag}_ab.err
python - <<'PY' $out/${tag}_ab.txt
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    if "variant" in d: print(" ", d["variant"]); continue
    if "us_per_call" in d: print("      rows", d["rows"], "len", d["prefix_len"], d["us_per_call"], "us", d["avg_allowed_tokens_first8rows"])
PY
tail -3 $out/${tag}_ab.err
