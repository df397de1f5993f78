#!/bin/bash
# build here (cross-compile), then run a command on a GPU box: tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
