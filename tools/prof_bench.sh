#!/bin/bash
# rocprofv3 passes over bench.py (profiles/README.md).  usage: tools/prof_bench.sh <outdir> <tag> [bench args...]
# pass 1: kernel trace + stats; then (separate runs, as the MI355X guide prescribes: one --pmc counter set per run, never with sys / hip traces):
# pass 2: FETCH_SIZE of the constraint kernels (k_constrain*, k_table_bits, k_beam_advance) -> <tag>_pmc_fetch_size.json (bench.py: roofline.traffic)
# pass 3 / 4: FETCH_SIZE / WRITE_SIZE of the aggregation kernels -> <tag>_pmc_agg.json (bench.py: roofline_aggregate.traffic)
out=$1; tag=$2; shift; shift
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_pmc /tmp/prof_aggf /tmp/prof_aggw
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
AGG="k_agg_locate|k_occ_prepare|k_mis|k_doc_keys|k_heads|k_entry_starts|k_entries|k_sel_|k_select_top|k_pad_entries|k_gather|k_top_docs|k_scatter|k_full_score|k_rank_docs|rocprim"
SEAL_BENCH_SKIP_OTHER=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $B "$@" > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_rocprof.log 2>&1
f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$out/${tag}_kernel_stats.csv
# (the full trace -- 30-60 MB -- stays on the box: gpurun_out/ travels back only below 64 MiB; KEEP_TRACE=1 copies it)
f=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/trace_by_category.py $f 3 > $GRAFT_REPO_ROOT/$out/${tag}_trace_by_category.txt 2>&1
[ -n "$f" ] && [ -n "$KEEP_TRACE" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_trace_full.csv
if [ -z "$SKIP_PMC" ]; then
SEAL_BENCH_SKIP_OTHER=1 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_expand|k_prefix_ranges|k_constrain|k_table_bits|k_beam_advance" --output-format csv -d /tmp/prof_pmc -- $B "$@" > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_pmc.log 2>&1
wt=$(grep -o "workload_tag=[^ ]*" $GRAFT_REPO_ROOT/$out/${tag}_bench_under_pmc.log | head -1 | cut -d= -f2)
f=$(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $f $wt > $GRAFT_REPO_ROOT/$out/${tag}_pmc_fetch_size.json
SEAL_BENCH_SKIP_OTHER=1 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$AGG" --output-format csv -d /tmp/prof_aggf -- $B "$@" > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_pmc_agg_fetch.log 2>&1
SEAL_BENCH_SKIP_OTHER=1 timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$AGG" --output-format csv -d /tmp/prof_aggw -- $B "$@" > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_pmc_agg_write.log 2>&1
ff=$(find /tmp/prof_aggf -name "*counter_collection.csv" | head -1); fw=$(find /tmp/prof_aggw -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python $GRAFT_REPO_ROOT/tools/summarize_pmc_agg.py $ff $fw $wt > $GRAFT_REPO_ROOT/$out/${tag}_pmc_agg.json
fi
ls -la $GRAFT_REPO_ROOT/$out | tail -12
