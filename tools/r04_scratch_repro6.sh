#!/bin/bash
# Round 4, item 1, sixth pass.  Fifth pass: with a private segment in pass 1 the pass-1 OUTPUT is already wrong right behind
# the kernel on its own stream - every tuple in its partition, some twice and some missing: slots that were never written
# (they keep the previous step's tuple of the same partition; in a step 0 they hold nothing and ~10^6 matches are lost).
# (1) does a wave that still has stores in flight when it ends lose them (variant 14 = variant 9 + s_waitcnt vmcnt(0) at the end)?
# (2) the same without any join code: tools/ubench_scratch_race.hip now stamps and verifies its output.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro6.txt
V=hash_join_codes_knl_amd/lib/variants
echo "# r04 scratch reproduction, sixth pass, $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT
one() {  # one <title> <lib> <steps> -- <args...>
  local title="$1" lib="$2" steps="$3"; shift 4
  { echo; echo "### $title"; echo "\$ HJGPU_DEBUG_FLAT_PRIORITIES=1 HJGPU_LIBRARY=$lib python tools/stress_cpra.py --steps $steps $*"
    env HJGPU_DEBUG_FLAT_PRIORITIES=1 HJGPU_LIBRARY=$PWD/$lib timeout -k 10 300 python tools/stress_cpra.py --steps $steps "$@" 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path\|WRONG: count" | cut -c1-300 | head -40
    echo "rc=${PIPESTATUS[0]}"; } >> $OUT 2>&1
}
one "variant 9 (private word in pass 1), setting A" $V/scratch_exp9.so 40 -- --option exchange_in_place=0
one "variant 14 (variant 9 + drained tail), setting A" $V/scratch_exp14.so 80 -- --option exchange_in_place=0
one "variant 14, setting A, forensics" $V/scratch_exp14.so 30 -- --option exchange_in_place=0 --option debug_forensics=1
{ echo; echo "## stand-alone (tools/ubench_scratch_race.hip)"; timeout -k 10 200 hash_join_codes_knl_amd/lib/ubench_scratch_race ${UB_SECONDS:-8} 2>&1; echo "ubench rc=$?"; } >> $OUT 2>&1
grep -E '^###|steps wrong|^mode|mode [0-9] =' $OUT | cut -c1-250
