#!/bin/bash
# Round 4, item 1, fifth pass: WHERE is a slice wrong?  Variant 9 (pass 1 carries a private word it never touches inside its
# loop) fails ~100 % in setting A (flat priorities, copying exchange); variant 13 = the product's kernels with the same
# forensic instrumentation (control).  Option debug_forensics=1 checks every probe slice behind pass 1 on the partitioning
# stream, in front of its join on the join stream, and (2 slices) in memory once the step is over.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro5.txt
V=hash_join_codes_knl_amd/lib/variants
echo "# r04 scratch reproduction, fifth pass, $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT
one() {  # one <title> <lib> <steps> -- <args...>
  local title="$1" lib="$2" steps="$3"; shift 4
  { echo; echo "### $title"; echo "\$ HJGPU_DEBUG_FLAT_PRIORITIES=1 HJGPU_LIBRARY=$lib python tools/stress_cpra.py --steps $steps $*"
    env HJGPU_DEBUG_FLAT_PRIORITIES=1 HJGPU_LIBRARY=$PWD/$lib timeout -k 10 300 python tools/stress_cpra.py --steps $steps "$@" 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" | head -150
    echo "rc=${PIPESTATUS[0]}"; } >> $OUT 2>&1
}
one "control (variant 13: product kernels), 2 slices, forensics" $V/scratch_exp13.so 30 -- --slices 2 --option exchange_in_place=0 --option debug_forensics=1
one "variant 9, 2 slices, no forensics (base rate)" $V/scratch_exp9.so 30 -- --slices 2 --option exchange_in_place=0
one "variant 9, 2 slices, forensics" $V/scratch_exp9.so 30 -- --slices 2 --option exchange_in_place=0 --option debug_forensics=1
one "variant 9, 8 slices, forensics" $V/scratch_exp9.so 20 -- --slices 8 --option exchange_in_place=0 --option debug_forensics=1
one "variant 12, 2 slices, forensics" $V/scratch_exp12.so 20 -- --slices 2 --option exchange_in_place=0 --option debug_forensics=1
tail -60 $OUT
