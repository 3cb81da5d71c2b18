#!/bin/bash
# Round 4, item 1, third pass: variant 3 (pass 1 with a private segment, on the low-priority partitioning stream) fails in
# a third of the pipeline's steps and never on one stream, with every re-read private value right.  Which knob moves it?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro3.txt
V=hash_join_codes_knl_amd/lib/variants
STEPS=${STEPS:-60}
LIB=${LIB:-$V/scratch_exp3.so}
echo "# r04 scratch reproduction, third pass ($LIB), $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT
one() {  # one <title> <env...> -- <args...>
  local title="$1"; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  { echo; echo "### $title"; echo "\$ ${envs[*]} HJGPU_LIBRARY=$LIB python tools/stress_cpra.py --steps $STEPS $*"
    env "${envs[@]}" HJGPU_LIBRARY=$PWD/$LIB timeout -k 10 300 python tools/stress_cpra.py --steps $STEPS "$@" 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path\|WRONG: count"
    echo "rc=${PIPESTATUS[0]}"; } >> $OUT 2>&1
}
one "baseline" --
one "AMD_OPT_FLUSH=0 (system-scope fences around every kernel)" AMD_OPT_FLUSH=0 --
one "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0 --
one "HSA_ENABLE_SCRATCH_ALT=0" HSA_ENABLE_SCRATCH_ALT=0 --
one "HSA_NO_SCRATCH_RECLAIM=1" HSA_NO_SCRATCH_RECLAIM=1 --
one "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 --
one "GPU_MAX_HW_QUEUES=1" GPU_MAX_HW_QUEUES=1 --
one "GPU_MAX_HW_QUEUES=2" GPU_MAX_HW_QUEUES=2 --
one "all four streams at the default priority" HJGPU_DEBUG_FLAT_PRIORITIES=1 --
one "debug_serialize=10 (partitioning and joins never overlap)" -- --option debug_serialize=10
one "debug_serialize=15 (+ host waits, exchange waits)" -- --option debug_serialize=15
one "AMD_SERIALIZE_KERNEL=3" AMD_SERIALIZE_KERNEL=3 --
one "exchange_in_place=0" -- --option exchange_in_place=0
one "2 slices" -- --slices 2
one "baseline again" --
grep -E '^###|steps wrong' $OUT
