#!/bin/sh
# build everything that travels (the HIP library, the oracle, oracle/_ref), then one gpurun call:   tools/gpu.sh <timeout s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
