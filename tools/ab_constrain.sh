#!/bin/bash
# A/B of k_constrain's two launch shapes on one box: tools/ab_constrain.sh <tag> [bench]
tag=$1
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_gputest.log 2>&1
echo "pytest rc=$?"; tail -5 $out/${tag}_gputest.log
V='SEALFM_CONSTRAIN_WAVES=8|SEALFM_CONSTRAIN_WAVES=1|SEALFM_CONSTRAIN_WAVES=8|SEALFM_CONSTRAIN_WAVES=1'
echo "== counters off, full prefix search" >> $out/${tag}_ab.txt
EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --prefix-len 1,2,3 --iters 20 --variants "$V" >> $out/${tag}_ab.txt 2>> $out/${tag}_ab.err
echo "== counters off, incremental (the decode step's call)" >> $out/${tag}_ab.txt
EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --prefix-len 2,3,4,6 --iters 20 --incremental --variants "$V" >> $out/${tag}_ab.txt 2>> $out/${tag}_ab.err
echo "== waves per workgroup: 8, counters on, stamps" >> $out/${tag}_ab.txt
timeout 600 python tools/expand_bench.py --prefix-len 1,3 --iters 20 --timestamps >> $out/${tag}_ab.txt 2>> $out/${tag}_ab.err
timeout 600 python tools/expand_bench.py --prefix-len 3,6 --iters 20 --timestamps --incremental >> $out/${tag}_ab.txt 2>> $out/${tag}_ab.err
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
  echo "bench rc=$?"; tail -3 $out/${tag}_bench.log; python - <<'PY' $out/${tag}_bench.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "achieved", "lane_pair_utilisation")}, d["parity_check"]["mismatches"], d["extra"]["phase_ms_one_batch"])
PY
fi
