#!/bin/bash
# Round 4, item 1, fourth pass.  Third pass: variant 3 fails only while pass 1 and the join stream's kernels overlap (0 of 60
# with debug_serialize=10 or AMD_SERIALIZE_KERNEL=3), at ~100 % with all streams at one priority or with the copying exchange;
# no scratch knob of the runtime matters.  Which ingredient of variant 3 is needed - and does the PRODUCT fail in that
# most aggressive setting?  Setting A = flat priorities + exchange_in_place=0.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro4.txt
V=hash_join_codes_knl_amd/lib/variants
echo "# r04 scratch reproduction, fourth pass, $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT
one() {  # one <title> <lib or ""> <steps> <env...> -- <args...>
  local title="$1" lib="$2" steps="$3"; shift 3
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  { echo; echo "### $title"; echo "\$ ${envs[*]} ${lib:+HJGPU_LIBRARY=$lib} python tools/stress_cpra.py --steps $steps $*"
    env "${envs[@]}" ${lib:+HJGPU_LIBRARY=$PWD/$lib} timeout -k 10 300 python tools/stress_cpra.py --steps $steps "$@" 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path\|WRONG: count"
    echo "rc=${PIPESTATUS[0]}"; } >> $OUT 2>&1
}
A="HJGPU_DEBUG_FLAT_PRIORITIES=1"
one "PRODUCT, setting A, 300 steps" "" 300 $A -- --option exchange_in_place=0
one "PRODUCT, flat priorities, in place, 300 steps" "" 300 $A --
one "PRODUCT, setting A, loopback world 2, 150 steps" "" 150 $A -- --option exchange_in_place=0 --transport loopback --world 2
for e in 3 9 10 11 12 1 5 6; do one "variant $e, setting A" $V/scratch_exp$e.so 40 $A -- --option exchange_in_place=0; done
grep -E '^###|steps wrong' $OUT
