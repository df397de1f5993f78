#!/bin/bash
# A/B of k_constrain's launch shapes on one box: tools/ab_constrain.sh <tag> [bench]
tag=$1
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_gputest.log 2>&1
echo "pytest rc=$?"; tail -5 $out/${tag}_gputest.log
V='SEALFM_CONSTRAIN_WAVES=8 SEALFM_CONSTRAIN_WAIT=0:0 SEALFM_CONSTRAIN_XCD=0'
V="$V|SEALFM_CONSTRAIN_WAVES=8 SEALFM_CONSTRAIN_WAIT=768:400 SEALFM_CONSTRAIN_XCD=0"
V="$V|SEALFM_CONSTRAIN_WAVES=8 SEALFM_CONSTRAIN_WAIT=768:800 SEALFM_CONSTRAIN_XCD=0"
V="$V|SEALFM_CONSTRAIN_WAVES=8 SEALFM_CONSTRAIN_WAIT=256:400 SEALFM_CONSTRAIN_XCD=0"
V="$V|SEALFM_CONSTRAIN_WAVES=8 SEALFM_CONSTRAIN_WAIT=0:0 SEALFM_CONSTRAIN_XCD=1"
V="$V|SEALFM_CONSTRAIN_WAVES=8 SEALFM_CONSTRAIN_WAIT=768:400 SEALFM_CONSTRAIN_XCD=1"
V="$V|SEALFM_CONSTRAIN_WAVES=16 SEALFM_CONSTRAIN_WAIT=0:0 SEALFM_CONSTRAIN_XCD=0"
V="$V|SEALFM_CONSTRAIN_WAVES=16 SEALFM_CONSTRAIN_WAIT=1536:400 SEALFM_CONSTRAIN_XCD=0"
V="$V|SEALFM_CONSTRAIN_WAVES=4 SEALFM_CONSTRAIN_WAIT=0:0 SEALFM_CONSTRAIN_XCD=0"
V="$V|SEALFM_CONSTRAIN_WAVES=4 SEALFM_CONSTRAIN_WAIT=384:400 SEALFM_CONSTRAIN_XCD=0"
V="$V|SEALFM_CONSTRAIN_WAVES=1"
echo "== counters off, full prefix search" >> $out/${tag}_ab.txt
EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --prefix-len 1,2,3 --iters 20 --variants "$V" >> $out/${tag}_ab.txt 2>> $out/${tag}_ab.err
echo "== counters off, incremental (the decode step's call)" >> $out/${tag}_ab.txt
EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --prefix-len 2,3,4,6 --iters 20 --incremental --variants "$V" >> $out/${tag}_ab.txt 2>> $out/${tag}_ab.err
echo "== waves per workgroup: 8, wait 768:400, counters on, stamps" >> $out/${tag}_ab.txt
SEALFM_CONSTRAIN_WAIT=768:400 timeout 600 python tools/expand_bench.py --prefix-len 1,2 --iters 20 --timestamps >> $out/${tag}_ab.txt 2>> $out/${tag}_ab.err
python - <<'PY' $out/${tag}_ab.txt
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("=="): print(line); continue
    d = json.loads(line)
    if "variant" in d: print(" ", d["variant"].replace("SEALFM_CONSTRAIN_", "")); continue
    if "us_per_call" in d: print("      len", d["prefix_len"], d["us_per_call"], "us", d.get("lane_pair_util"), d["avg_allowed_tokens_first8rows"])
    else: print("     ", line)
PY
tail -3 $out/${tag}_ab.err
if [ -n "$2" ]; then
  timeout 1200 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.log
  echo "bench rc=$?"; tail -3 $out/${tag}_bench.log; cat $out/${tag}_bench.json
fi
