#!/bin/bash
# Round 5, item 1: plain against non-temporal K6 stores where wrong steps are most frequent (all streams at one priority: 3 wrong
# steps in 10 000, profiles/r05_coresidency.txt).  usage (GPU box): bash tools/r05_store_policy2.sh -> gpurun_out/r05_store_policy2.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export HJGPU_DEBUG_FLAT_PRIORITIES=1
OUT=gpurun_out/r05_store_policy2.txt
mkdir -p gpurun_out
echo "# tools/r05_store_policy2.sh (HJGPU_DEBUG_FLAT_PRIORITIES=1), $(date -u +%FT%RZ)" > $OUT
run() { echo "## HJGPU_LIBRARY=${HJGPU_LIBRARY:-product} $*" >> $OUT; timeout -k 10 700 python3 "$@" 2>&1 | grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" >> $OUT; }
run tools/stress_cpra.py --steps ${1:-15000} --slices 8
export HJGPU_LIBRARY=$PWD/hash_join_codes_knl_amd/lib/variants/k6nt.so
run tools/stress_cpra.py --steps ${2:-45000} --slices 8
cat $OUT | grep -v "^\.\.\."
