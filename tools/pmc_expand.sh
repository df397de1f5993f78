#!/bin/bash
# PMC passes over tools/expand_bench.py (k_constrain).  usage: tools/pmc_expand.sh <outdir> [expand_bench args...]
# One counter group per pass (rocprofv3 --pmc with --kernel-trace only, as gpurun requires).
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "FETCH_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  d=/tmp/pmc_$i
  rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "k_constrain|k_expand" --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/expand_bench.py "$@" > $d.log 2>&1)
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $GRAFT_REPO_ROOT/tools/summarize_pmc.py $f > "$out/pmc_$i.json"; else echo "pass $i failed"; tail -5 $d.log; fi
done
