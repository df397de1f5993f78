#!/bin/bash
# pipelining variants of the search step on one box (same index build each: ~25 s per run)
out=gpurun_out; mkdir -p $out
for v in "SEAL_OVERLAP_DEPTH=1" "SEAL_OVERLAP_DEPTH=2" "SEAL_OVERLAP_DEPTH=1"; do
  env $v SEAL_BENCH_SKIP_OTHER=1 timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $out/ov.json 2> $out/ov.log
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ov.json").read().strip().splitlines()[-1])
print(sys.argv[1], "->", d["value"], "q/s", d["ms_per_step"], "ms")
PY
done
