#!/bin/bash
# Round 5: what non-temporal stores cost K6 (pass 1 / pass 2, same allocations for all builds), whether written-through stores
# (sc0 sc1) keep every tuple too, and the host-column multi-GPU tests after the CPRA upload change.
# usage (GPU box): bash tools/r05_ab_stores.sh -> gpurun_out/r05_ab_stores.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05_ab_stores.txt
V=hash_join_codes_knl_amd/lib/variants
mkdir -p gpurun_out
echo "# tools/r05_ab_stores.sh, $(date -u +%FT%RZ)" > $OUT
echo "## ab_libs --sequential: product (nt), nt 16-byte stores only, nt 8-byte stores only, plain" >> $OUT
timeout -k 10 400 python3 tools/ab_libs.py hash_join_codes_knl_amd/lib/libhjgpu.so $V/k6nt16.so $V/k6nt8.so $V/k6plain.so --sequential --rounds 4 --reps 3 2>&1 | grep -v amdgpu.ids >> $OUT
echo "## pytest tests/test_gpu_multi.py -k host" >> $OUT
timeout -k 10 500 python3 -m pytest tests/test_gpu_multi.py -m gpu -q -x -k "host" 2>&1 | tail -3 >> $OUT
echo "## HJGPU_LIBRARY=variants/k6wt2.so HJGPU_DEBUG_FLAT_PRIORITIES=1 stress_cpra.py --steps 15000 --slices 8 (system-scope, written-through K6 stores)" >> $OUT
HJGPU_LIBRARY=$PWD/$V/k6wt2.so HJGPU_DEBUG_FLAT_PRIORITIES=1 timeout -k 10 500 python3 tools/stress_cpra.py --steps 15000 --slices 8 2>&1 | grep --line-buffered -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tee -a $OUT | grep --line-buffered steps
grep -v "^\.\.\." $OUT
