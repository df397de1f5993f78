#!/bin/bash
# one GPU session: parity tests, the isolated constraint-kernel measurement, the bench line and (if the
# tests are green) the rocprofv3 passes.  usage: tools/gpu_round.sh <tag> [skip-prof]
tag=$1
out=gpurun_out
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/${tag}_gputest.log 2>&1
rc=$?
tail -25 $out/${tag}_gputest.log
echo "pytest rc=$rc"
timeout 600 python tools/expand_bench.py --prefix-len 1,2,3,5 --iters 20 > $out/${tag}_expand.txt 2> $out/${tag}_expand.err
EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --prefix-len 1,2,3,5 --iters 20 > $out/${tag}_expand_nocount.txt 2>> $out/${tag}_expand.err
cat $out/${tag}_expand.txt $out/${tag}_expand_nocount.txt; tail -3 $out/${tag}_expand.err
timeout 1200 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.log
echo "bench rc=$?"; tail -5 $out/${tag}_bench.log; cat $out/${tag}_bench.json
if [ $rc -eq 0 ] && [ -z "$2" ]; then
  bash tools/prof_bench.sh $out $tag
fi
