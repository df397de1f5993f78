#!/bin/bash
# round 4, call 16: configs[4] (stress) with its own PMC pass, then its line once more (is call 14's 247 queries/s the box or the code?)
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_pmc_s && SEAL_BENCH_SKIP_OTHER=1 timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_constrain|k_table_bits" --output-format csv -d /tmp/prof_pmc_s -- python $GRAFT_REPO_ROOT/bench.py --workload stress --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/r4_stress_bench_under_pmc.log 2>&1 )
wt=$(grep -o "workload_tag=[^ ]*" $out/r4_stress_bench_under_pmc.log | head -1 | cut -d= -f2)
f=$(find /tmp/prof_pmc_s -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/summarize_pmc.py $f $wt > $out/r4_pmc_fetch_size_stress.json && cp $out/r4_pmc_fetch_size_stress.json profiles/
echo "stress pmc: $wt $(ls $out/r4_pmc_fetch_size_stress.json 2>/dev/null | wc -l)"
timeout -s ABRT 700 python -X faulthandler bench.py --workload stress --steps 5 --warmup 2 > $out/r4_stress_bench_b.json 2> $out/r4_stress_bench_b.log; echo "stress rc=$?"
python - <<'PY' $out/r4_stress_bench_b.json
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc = d.get("parity_check") or {}
    print({k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "traffic", "algorithmic_bytes_per_launch")}, "mismatches", pc.get("mismatches"), pc.get("values_compared"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    print("   ", d["extra"].get("phase_ms_one_batch"), d["extra"].get("p50_batch_latency_ms_unpipelined"), d["extra"].get("prefix_tables"), (d["roofline"].get("traffic_source") or {}).get("file"))
except Exception as e:
    print("no line:", e)
PY
