#!/bin/bash
# round 4, call 11: k_constrain_table with symbol-space byte stores + k_table_bits: parity of the constraint forms, A/B with grids
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_fmindex.py tests/test_gpu_decode.py -m gpu -x -q > $out/r4_gputest_tables2.log 2>&1; echo "pytest rc=$?"; tail -3 $out/r4_gputest_tables2.log
V="SEALFM_PREFIX_TABLES=0|SEALFM_PREFIX_TABLES=1 SEALFM_TABLE_GRID=1024|SEALFM_PREFIX_TABLES=1 SEALFM_TABLE_GRID=512|SEALFM_PREFIX_TABLES=1 SEALFM_TABLE_GRID=2048|SEALFM_PREFIX_TABLES=1 SEALFM_TABLE_GRID=4096"
EXPAND_NO_COUNT=1 timeout 300 python tools/expand_bench.py --docs 21015324 --rows 600 --prefix-len 1 --iters 30 --variants "$V" > $out/r4_tables2_ab_600.txt 2>&1; echo "ab 600 rc=$?"
grep -h "variant\|us_per_call" $out/r4_tables2_ab_600.txt | cut -c1-60,150-200
