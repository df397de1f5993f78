#!/bin/bash
# round 4, first GPU call: the split GEMM (seal_amd/split_gemm.py) -- probe, its GPU test, then the bench line with it on / off.
# tools/r4_split_gemm.sh <tag>
tag=${1:-r4s}
out=gpurun_out; mkdir -p $out
timeout 200 python tools/split_gemm_probe.py > $out/${tag}_probe.txt 2>&1; echo "probe rc=$?"; grep -v "^{" $out/${tag}_probe.txt | tail -12
# the same product on hipBLASLt directly (every algorithm the heuristic offers): the route if torch's out_dtype path is missing or slow
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/blaslt_f16_probe.cpp -lhipblaslt -o /tmp/blaslt_probe > $out/${tag}_blaslt_build.log 2>&1 \
  && timeout 200 /tmp/blaslt_probe > $out/${tag}_blaslt.txt 2>&1; echo "blaslt probe rc=$?"; tail -22 $out/${tag}_blaslt.txt
SEAL_TEST_SPLIT_GEMM=1 timeout 200 python -m pytest tests/test_split_gemm.py -x -q > $out/${tag}_test.log 2>&1; echo "test rc=$?"; tail -3 $out/${tag}_test.log
# GEMM changes do not depend on the index: the A/B runs on a 2 M-passage corpus without the CPU legs (~30 s per leg instead of ~90 s;
# score parity against HF's forward is still in the log); FULL=1 runs the default line instead
QUICK="--docs 2000000 --corpus-phrases 2000000 --no-cpu-baseline"
[ -n "$FULL" ] && QUICK=""
for mode in 1 0; do
  # (SEAL_SPLIT_GEMM_MIN_N / _MIN_ROWS: set from the probe's per-shape times before this leg)
  SEAL_BENCH_SCORE_PARITY=1 SEAL_SPLIT_GEMM=$mode timeout -s ABRT 300 python -X faulthandler bench.py $QUICK --steps 20 --warmup 5 > $out/${tag}_bench_split$mode.json 2> $out/${tag}_bench_split$mode.log
  echo "bench(split gemm $mode) rc=$?"
  python - <<'PY' $out/${tag}_bench_split$mode.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pc = d.get("parity_check") or {}
print({k: d[k] for k in ("value", "ms_per_step")}, "mismatches", pc.get("mismatches"), d["extra"].get("phase_ms_one_batch"))
PY
  grep "score parity" $out/${tag}_bench_split$mode.log | cut -c1-250
done
