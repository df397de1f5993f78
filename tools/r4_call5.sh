#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 120 python tools/debug_gelu_planes.py > $out/r4c5_gelu.txt 2>&1; tail -15 $out/r4c5_gelu.txt | cut -c1-250
python tools/soak_ctl.py excl2 > $out/r4c5_excl2.txt 2>&1; grep "^==\|CLEAN\|STALL\|differs" $out/r4c5_excl2.txt | cut -c1-230
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/r4c5_gputest.log 2>&1; echo "gpu tests rc=$?"; grep -n "^FAILED\|^ERROR\|passed\|failed" $out/r4c5_gputest.log | tail -12 | cut -c1-250
