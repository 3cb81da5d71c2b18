#!/bin/bash
# Round 4, item 1, second pass: the first pass (gpurun_out/r04_scratch_repro.txt) showed the private segment returning every
# value it was given (0 mismatches in 10^10 re-reads) while the variants that USE it in pass 2 / pass 1 gave wrong joins.
# Is that tied to other streams at all?  Every variant on ONE stream, then the pipeline fully serialised.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro2.txt
V=hash_join_codes_knl_amd/lib/variants
STEPS=${STEPS:-40}
echo "# r04 scratch reproduction, second pass, $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT
one() {  # one <title> <lib or ""> <cmd...>
  local title="$1" lib="$2"; shift 2
  { echo; echo "### $title"; echo "\$ ${lib:+HJGPU_LIBRARY=$lib }$*"
    env ${lib:+HJGPU_LIBRARY=$PWD/$lib} timeout -k 10 300 "$@" 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path"
    echo "rc=${PIPESTATUS[0]}"; } >> $OUT 2>&1
}
echo "## one GPU, ONE stream: hjgpu_phj 64 M x 1 G, $STEPS checked steps" >> $OUT
one "product" "" python tools/stress_single.py --steps $STEPS
for e in 1 2 3 5 6 7 8; do one "variant $e" $V/scratch_exp$e.so python tools/stress_single.py --steps $STEPS; done
one "variant 2, one-GPU CPRA (8 chunks)" $V/scratch_exp2.so python tools/stress_single.py --steps $STEPS --algo cpra
echo >> $OUT; echo "## the slice pipeline (RCCL world 1, 8 slices), 60 checked steps" >> $OUT
for e in 5 6 7 8; do one "variant $e" $V/scratch_exp$e.so python tools/stress_cpra.py --steps 60; done
one "variant 2, debug_serialize=7 (host waits after every join; partitioning and exchange wait for the joins)" $V/scratch_exp2.so python tools/stress_cpra.py --steps 60 --option debug_serialize=7
one "variant 2, one slice" $V/scratch_exp2.so python tools/stress_cpra.py --steps 60 --slices 1
one "variant 3, debug_serialize=7" $V/scratch_exp3.so python tools/stress_cpra.py --steps 60 --option debug_serialize=7
grep -E '^###|steps wrong|MISMATCH' $OUT | tail -60
