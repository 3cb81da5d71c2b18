#!/bin/bash
# Round 4, item 1, seventh pass: are pass 1's waves ever switched out (context save / restore) while other queues run?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro7.txt
V=hash_join_codes_knl_amd/lib/variants
echo "# r04 scratch reproduction, seventh pass, $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT
one() {  # one <title> <lib> -- <cmd...>
  local title="$1" lib="$2"; shift 3
  { echo; echo "### $title"; echo "\$ HJGPU_DEBUG_FLAT_PRIORITIES=1 HJGPU_LIBRARY=$lib $*"
    env HJGPU_DEBUG_FLAT_PRIORITIES=1 HJGPU_LIBRARY=$PWD/$lib timeout -k 10 300 "$@" 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path\|WRONG: count" | cut -c1-300 | head -40
    echo "rc=${PIPESTATUS[0]}"; } >> $OUT 2>&1
}
one "variant 16 (no private word, gaps), one stream" $V/scratch_exp16.so -- python tools/stress_single.py --steps 20
one "variant 15 (private word, gaps), one stream" $V/scratch_exp15.so -- python tools/stress_single.py --steps 20
one "variant 16 (no private word, gaps), pipeline, setting A" $V/scratch_exp16.so -- python tools/stress_cpra.py --steps 40 --option exchange_in_place=0
one "variant 15 (private word, gaps), pipeline, setting A" $V/scratch_exp15.so -- python tools/stress_cpra.py --steps 40 --option exchange_in_place=0
grep -E '^###|steps wrong|time between' $OUT | cut -c1-250
