#!/bin/bash
# Round 5, item 1: does K6 still lose stores next to other streams' kernels when its stores are written through (sc0 sc1) or
# non-temporal (nt)?  Library variants from tools/build_variant.py k6wt -DHJ_K6_STORE=2 / k6nt -DHJ_K6_STORE=1.
# usage (GPU box): bash tools/r05_store_policy.sh [steps] [variants...]  -> gpurun_out/r05_store_policy.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-28000}
shift || true
VARIANTS=${*:-k6wt k6nt}
OUT=gpurun_out/r05_store_policy.txt
mkdir -p gpurun_out
echo "# tools/r05_store_policy.sh $N $VARIANTS, $(date -u +%FT%RZ)" >> $OUT
for v in $VARIANTS; do
    echo "## HJGPU_LIBRARY=variants/$v.so stress_cpra.py --steps $N --slices 8" >> $OUT
    if [ "$v" = product ]; then unset HJGPU_LIBRARY; else export HJGPU_LIBRARY=$PWD/hash_join_codes_knl_amd/lib/variants/$v.so; fi
    timeout -k 10 500 python3 tools/stress_cpra.py --steps $N --slices 8 2>&1 | grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" >> $OUT
done
cat $OUT
