#!/bin/bash
# round 4, call 12: k_constrain_table with the next node prefetched behind the block load, batched row pass, gated superblock rows
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_fmindex.py tests/test_gpu_decode.py tests/test_gpu_search.py -m gpu -x -q > $out/r4_gputest_tables3.log 2>&1; echo "pytest rc=$?"; tail -3 $out/r4_gputest_tables3.log
V="SEALFM_PREFIX_TABLES=0|SEALFM_PREFIX_TABLES=1 SEALFM_TABLE_GRID=1024|SEALFM_PREFIX_TABLES=1 SEALFM_TABLE_GRID=2048|SEALFM_PREFIX_TABLES=1 SEALFM_TABLE_GRID=768"
for rows in 600 300; do
  EXPAND_NO_COUNT=1 timeout 300 python tools/expand_bench.py --docs 21015324 --rows $rows --prefix-len 1 --iters 30 --variants "$V" > $out/r4_tables3_ab_$rows.txt 2>&1; echo "ab $rows rc=$?"
done
timeout 300 python tools/expand_bench.py --docs 21015324 --rows 600 --prefix-len 1 --iters 30 --variants "SEALFM_PREFIX_TABLES=0|SEALFM_PREFIX_TABLES=1 SEALFM_TABLE_GRID=1024" > $out/r4_tables3_ab_600_counted.txt 2>&1
EXPAND_NO_COUNT=1 timeout 400 python tools/expand_bench.py --docs 36000000 --rows 600 --prefix-len 1 --iters 20 --variants "SEALFM_PREFIX_TABLES=0|SEALFM_PREFIX_TABLES=1 SEALFM_TABLE_GRID=1024" > $out/r4_tables3_ab_600_kilt.txt 2>&1; echo "kilt ab rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4_tables3_ab_*.txt')):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if 'variant' in d: print('  ', d['variant'])
        else: print('      ', d['us_per_call'], 'us', d['alg_MB_per_call'], 'MB', d['frac_of_8TBps'], d['bitmap_checksum'])
PY
