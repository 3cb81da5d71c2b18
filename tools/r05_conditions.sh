#!/bin/bash
# Round 5, item 1: under which conditions does the slice pipeline lose tuples (rate ~2e-4 per step in tools/r05_rate.sh)?
# usage (GPU box): bash tools/r05_conditions.sh  -> gpurun_out/r05_conditions.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05_conditions.txt
mkdir -p gpurun_out
echo "# tools/r05_conditions.sh, $(date -u +%FT%RZ)" > $OUT
run() { echo "## $*" >> $OUT; timeout -k 10 420 python3 "$@" 2>&1 | grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" >> $OUT; }
run tools/stress_single.py --algo phj --steps 15000
run tools/stress_cpra.py --steps 15000 --slices 8 --option debug_serialize=10
run tools/stress_cpra.py --steps 12000 --slices 1
run tools/stress_cpra.py --steps 15000 --slices 8 --option cpra_fused_counts=0
cat $OUT
