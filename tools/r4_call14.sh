#!/bin/bash
# round 4, call 14: the two larger tiers on the final kernels (prefix tables in): configs[3] (KILT size) with its own PMC pass, configs[4] (stress)
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_pmc_k && SEAL_BENCH_SKIP_OTHER=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_constrain|k_table_bits" --output-format csv -d /tmp/prof_pmc_k -- python $GRAFT_REPO_ROOT/bench.py --docs 36000000 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/r4_kilt_bench_under_pmc.log 2>&1 )
wt=$(grep -o "workload_tag=[^ ]*" $out/r4_kilt_bench_under_pmc.log | head -1 | cut -d= -f2)
f=$(find /tmp/prof_pmc_k -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/summarize_pmc.py $f $wt > $out/r4_pmc_fetch_size_kilt.json && cp $out/r4_pmc_fetch_size_kilt.json profiles/
echo "kilt pmc: $wt $(ls $out/r4_pmc_fetch_size_kilt.json 2>/dev/null | wc -l)"
timeout -s ABRT 700 python -X faulthandler bench.py --docs 36000000 --steps 5 --warmup 2 --cpu-locate-sample 16 > $out/r4_kilt_bench.json 2> $out/r4_kilt_bench.log; echo "kilt rc=$?"
timeout -s ABRT 840 python -X faulthandler bench.py --workload stress --steps 5 --warmup 2 > $out/r4_stress_bench.json 2> $out/r4_stress_bench.log; echo "stress rc=$?"
for f in $out/r4_kilt_bench.json $out/r4_stress_bench.json; do python - <<'PY' $f
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc = d.get("parity_check") or {}
    print(sys.argv[1].split("/")[-1], {k: d[k] for k in ("value", "ms_per_step")}, {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "traffic", "algorithmic_bytes_per_launch")}, "mismatches", pc.get("mismatches"), pc.get("values_compared"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d["config"]["index_hbm_gib"])
    print("   ", d["extra"].get("phase_ms_one_batch"), (d["roofline"].get("traffic_source") or {}).get("file"))
except Exception as e:
    print("no line:", sys.argv[1], e)
PY
done
grep "\[bench\]" $out/r4_stress_bench.log | cut -c1-200 | tail -12
