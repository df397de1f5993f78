#!/bin/bash
# PMC passes over the hand-written product (k_hgemm_nt / k_hgemm_tall) at a decode step's shapes: one counter group per pass (rocprofv3 --pmc with
# --kernel-trace only).  usage: tools/pmc_hgemm.sh <outfile> [rows]
out=$1; rows=${2:-600}
export TMPDIR=/tmp
: > "$out"
for shape in dxd qkv fc1 fc2 lm_head; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
             "FETCH_SIZE"; do      # (round 6: "TCC_HIT_sum TCC_MISS_sum FETCH_SIZE" in one pass aborted inside rocprofv3 on this pool, after its 200 s timeout: 17 GPU-minutes for five shapes)
    i=$((i+1)); d=/tmp/pmc_hg_${shape}_$i; rm -rf $d
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "k_hgemm" --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/hgemm_pmc_driver.py $shape $rows > $d.log 2>&1)
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then echo "== $shape rows $rows: $grp" >> "$out"; python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $f >> "$out"; else echo "== $shape pass $i failed" >> "$out"; tail -3 $d.log >> "$out"; fi
  done
done
