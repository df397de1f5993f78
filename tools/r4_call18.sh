#!/bin/bash
# round 4, call 18 (last): soak of the final code (60 repetitions of the timed call, --check), rocprofv3 kernel stats of the final model side
out=gpurun_out; mkdir -p $out
python tools/soak_ctl.py final > $out/r4_soak_final.txt 2>&1; grep "^==\|CLEAN\|STALL\|differs\|median" $out/r4_soak_final.txt | cut -c1-230 | tail -8
SKIP_PMC=1 bash tools/prof_bench.sh $out r4b > $out/r4b_prof_bench.log 2>&1; echo "prof rc=$?"; ls $out | grep "^r4b_"
