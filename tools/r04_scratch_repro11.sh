#!/bin/bash
# Round 4, item 1, which kernels: the private word in K4 (variant 17), in the join kernel (18), in pass 2 (19), in pass 1 (9) - each
# through tools/scratch_two_streams.py (partitioning on one stream, whole joins of the same library on another); and variant 9
# with K6's other stream-out path (--unpacked: separate key / payload columns).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro11.txt
V=hash_join_codes_knl_amd/lib/variants
echo "# r04 scratch reproduction, which kernels (tools/scratch_two_streams.py), $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT
for run in "9 " "9 --unpacked" "17 " "18 " "19 " "13 "; do
  set -- $run
  { echo; echo "### variant $1 $2"
    HJGPU_LIBRARY=$PWD/$V/scratch_exp$1.so timeout -k 10 200 python tools/scratch_two_streams.py --steps 40 $2 2>&1 | grep -v amdgpu.ids | grep -v "runs of unwritten" | cut -c1-220 | tail -3; } >> $OUT 2>&1
done
cat $OUT
