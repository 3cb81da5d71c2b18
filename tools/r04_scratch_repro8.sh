#!/bin/bash
# Round 4, item 1, eighth pass: the same effect WITHOUT the slice pipeline - two independent library calls on two streams.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro8.txt
V=hash_join_codes_knl_amd/lib/variants
echo "# r04 scratch reproduction, eighth pass (tools/scratch_two_streams.py), $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT
one() {  # one <title> <lib or ""> -- <args...>
  local title="$1" lib="$2"; shift 3
  { echo; echo "### $title"; echo "\$ ${lib:+HJGPU_LIBRARY=$lib }python tools/scratch_two_streams.py $*"
    env ${lib:+HJGPU_LIBRARY=$PWD/$lib} timeout -k 10 300 python tools/scratch_two_streams.py "$@" 2>&1 | grep -v "amdgpu.ids" | cut -c1-300 | head -20
    echo "rc=${PIPESTATUS[0]}"; } >> $OUT 2>&1
}
one "product library, joins next to the partitioning" "" -- --steps 60
one "variant 9 (pass 1 carries one private word), NOTHING next to the partitioning" $V/scratch_exp9.so -- --steps 40 --neighbour 0
one "variant 9, joins next to the partitioning" $V/scratch_exp9.so -- --steps 40
one "variant 13 (product kernels, control build), joins next to the partitioning" $V/scratch_exp13.so -- --steps 40
cat $OUT | tail -40
