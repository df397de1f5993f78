#!/bin/bash
# rocprofv3 passes over bench.py (profiles/README.md).  usage: tools/prof_bench.sh <outdir> <tag> [bench args...]
# pass 1: kernel trace + stats; pass 2 (separate run, as the MI355X guide prescribes): PMC FETCH_SIZE of the constraint kernels
out=$1; tag=$2; shift; shift
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_pmc
SEAL_BENCH_SKIP_OTHER=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_rocprof.log 2>&1
f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$out/${tag}_kernel_stats.csv
f=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_trace_full.csv
if [ -z "$SKIP_PMC" ]; then
SEAL_BENCH_SKIP_OTHER=1 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_expand|k_prefix_ranges|k_constrain|k_table_bits" --output-format csv -d /tmp/prof_pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_pmc.log 2>&1
wt=$(grep -o "workload_tag=[^ ]*" $GRAFT_REPO_ROOT/$out/${tag}_bench_under_pmc.log | head -1 | cut -d= -f2)
f=$(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $f $wt > $GRAFT_REPO_ROOT/$out/${tag}_pmc_fetch_size.json
fi
ls -la $GRAFT_REPO_ROOT/$out
