#!/bin/bash
# round 4, final evidence on the final code: GPU tests, rocprofv3 kernel stats + PMC pass of bench.py, the driver's bench line, the wide constraint call, soak
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/r4_gputest_final.log 2>&1; echo "gpu tests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed" $out/r4_gputest_final.log | tail -12 | cut -c1-250
bash tools/prof_bench.sh $out r4 > $out/r4_prof_bench.log 2>&1; echo "prof rc=$?"; ls $out | grep "^r4_" | head -20
cp $out/r4_pmc_fetch_size.json profiles/ 2>/dev/null
timeout -s ABRT 600 python -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $out/r4_bench_final.json 2> $out/r4_bench_final.log; echo "bench rc=$?"
grep "\[bench\]" $out/r4_bench_final.log | cut -c1-250 | tail -8
python - <<'PY' $out/r4_bench_final.json
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc = d.get("parity_check") or {}
    print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "launches", "traffic", "algorithmic_bytes_per_launch")}, "mismatches", pc.get("mismatches"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    print(d["extra"].get("phase_ms_one_batch"), d["extra"].get("p50_batch_latency_ms_unpipelined"))
except Exception as e:
    print("no bench line:", e)
PY
for rows in 300 600; do timeout 200 python tools/expand_bench.py --rows $rows --prefix-len 1 --iters 20 >> $out/r4_expand_bench.txt 2>> $out/r4_expand_bench.err; done
for rows in 300 600; do EXPAND_NO_COUNT=1 timeout 200 python tools/expand_bench.py --rows $rows --prefix-len 1,2,3,4,6 --iters 20 --incremental >> $out/r4_expand_bench_nocount.txt 2>> $out/r4_expand_bench.err; done
python - <<'PY' $out/r4_expand_bench.txt $out/r4_expand_bench_nocount.txt
import json, sys
for f in sys.argv[1:]:
    for line in open(f):
        try: d = json.loads(line)
        except Exception: continue
        if "us_per_call" in d: print("  ", f.split("/")[-1], "rows", d["rows"], "len", d["prefix_len"], d["us_per_call"], "us", d.get("alg_MB_per_call"), "MB", d.get("frac_of_8TBps"))
PY
python tools/soak_ctl.py soak > $out/r4_soak_final.txt 2>&1; grep "^==\|CLEAN\|STALL\|differs" $out/r4_soak_final.txt | cut -c1-230
