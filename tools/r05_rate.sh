#!/bin/bash
# Round 5, item 1a: the rate of wrong steps of the slice pipeline, default and _UNIQUE, on the library as round 4 shipped it.
# usage (GPU box): bash tools/r05_rate.sh [steps]   -> gpurun_out/r05_rate.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-10000}
OUT=gpurun_out/r05_rate.txt
mkdir -p gpurun_out
echo "# tools/r05_rate.sh $N, $(date -u +%FT%RZ)" > $OUT
rc=0
timeout -k 10 400 python3 tools/stress_cpra.py --steps $N --slices 8 --unique >> $OUT 2>&1 || rc=1
timeout -k 10 400 python3 tools/stress_cpra.py --steps $N --slices 8 >> $OUT 2>&1 || rc=1
timeout -k 10 300 python3 tools/stress_cpra.py --steps $((N / 4)) --slices 8 --world 2 --transport loopback --unique >> $OUT 2>&1 || rc=1
cat $OUT
exit $rc
