#!/bin/bash
# Round 5, item 1: (a) which stage's output lacks the tuples of a wrong step (stress_cpra.py --forensics: option "audit" on every
# context); (b) is a kernel that shares its CU with the fused K4 (74.5 KB of LDS, ONE workgroup per CU: a join or K4p workgroup of the
# other stream fits beside it) what loses them?  HJGPU_HIST_MIN_LDS makes K4 ask for 100 KB: nothing with a large LDS share fits.
# usage (GPU box): bash tools/r05_coresidency.sh  -> gpurun_out/r05_coresidency.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05_coresidency.txt
mkdir -p gpurun_out
echo "# tools/r05_coresidency.sh, $(date -u +%FT%RZ)" > $OUT
run() { echo "## $*" >> $OUT; timeout -k 10 ${LIMIT:-420} "$@" 2>&1 | grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" >> $OUT; }
run python3 tools/stress_cpra.py --steps 30 --slices 8 --forensics
grep -q "every stage's checksums agree" $OUT || { echo "forensics of a good step do not agree: stopping" >> $OUT; cat $OUT; exit 1; }
LIMIT=500 run python3 tools/stress_cpra.py --steps 12000 --slices 8 --forensics
run env HJGPU_HIST_MIN_LDS=102400 python3 tools/stress_cpra.py --steps 15000 --slices 8
run env HJGPU_DEBUG_FLAT_PRIORITIES=1 python3 tools/stress_cpra.py --steps 10000 --slices 8
cat $OUT
