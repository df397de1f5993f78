#!/bin/bash
# rocprofv3 passes over bench.py (profiles/README.md).  usage: tools/prof_bench.sh <outdir> [bench args...]
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_pmc
SEAL_BENCH_SKIP_OTHER=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$out/r1_bench_under_rocprof.log 2>&1
f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$out/r1_kernel_stats.csv
f=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && grep -E "Kernel_Name|k_expand|k_prefix|k_row|k_query|k_locate|k_get|k_bs|k_dense|k_apply|k_extract" $f > $GRAFT_REPO_ROOT/$out/r1_kernel_trace_index.csv
SEAL_BENCH_SKIP_OTHER=1 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_expand|k_prefix_ranges" --output-format csv -d /tmp/prof_pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$out/r1_bench_under_pmc.log 2>&1
f=$(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $f > $GRAFT_REPO_ROOT/$out/r1_pmc_fetch_size.json
ls -la $GRAFT_REPO_ROOT/$out
