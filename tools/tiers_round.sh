#!/bin/bash
# the larger tiers on one box: scale_check at 2 M and KILT size, k_constrain at KILT size and on the 1.4e10-symbol tier
tag=$1
out=gpurun_out
mkdir -p $out
{
echo '$ python tools/scale_check.py --docs 2000000'
timeout 600 python tools/scale_check.py --docs 2000000
echo '$ python tools/scale_check.py --docs 36000000'
timeout 900 python tools/scale_check.py --docs 36000000
} > $out/${tag}_scale_check.txt 2> $out/${tag}_scale_check.err
{
echo '$ python tools/expand_bench.py --docs 36000000 --prefix-len 1,2 --iters 10'
timeout 900 python tools/expand_bench.py --docs 36000000 --prefix-len 1,2 --iters 10
echo '$ python tools/expand_bench.py --synthetic-bwt 1.4e10 --rows 600 --prefix-len 1,2 --iters 10'
timeout 900 python tools/expand_bench.py --synthetic-bwt 1.4e10 --rows 600 --prefix-len 1,2 --iters 10
echo '$ python tools/expand_bench.py --prefix-len 3,4,6,8 --iters 20 --incremental   (counters on)'
timeout 600 python tools/expand_bench.py --prefix-len 3,4,6,8 --iters 20 --incremental
echo '$ EXPAND_NO_COUNT=1 python tools/expand_bench.py --prefix-len 3,4,6,8 --iters 20 --incremental'
EXPAND_NO_COUNT=1 timeout 600 python tools/expand_bench.py --prefix-len 3,4,6,8 --iters 20 --incremental
} > $out/${tag}_tiers.txt 2> $out/${tag}_tiers.err
cat $out/${tag}_scale_check.txt | cut -c1-400; tail -2 $out/${tag}_scale_check.err; cat $out/${tag}_tiers.txt | cut -c1-330; tail -2 $out/${tag}_tiers.err
