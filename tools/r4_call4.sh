#!/bin/bash
# round 4, GPU call 4: one library GEMM stream at a time -- soak with the split GEMM on, A/B, then the GPU tests
out=gpurun_out; mkdir -p $out
python tools/soak_ctl.py excl > $out/r4c4_excl.txt 2>&1; grep "^==\|CLEAN\|STALL\|differs" $out/r4c4_excl.txt | cut -c1-230
timeout 500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/r4c4_gputest.log 2>&1; echo "gpu tests rc=$?"; tail -8 $out/r4c4_gputest.log | cut -c1-300
