#!/bin/bash
# Round 4, item 1, last pass: does the loss depend on pass 1's geometry (LDS per workgroup, threads)?  tools/scratch_two_streams.py
# with variant 9 (one private word in pass 1) and the partitioning context's scatter_cfg = block,vectors,carry.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro10.txt
V=hash_join_codes_knl_amd/lib/variants
echo "# r04 scratch reproduction, geometry pass (tools/scratch_two_streams.py, variant 9), $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT
for cfg in "1024,4,1" "1024,3,1" "1024,2,1" "512,4,1" "512,2,1" "256,4,1" "256,2,1" "1024,4,0"; do
  { echo; echo "### scatter_cfg=$cfg (tile = block x vectors x 4 tuples; LDS ~ 8 bytes per tuple + carry)"
    HJGPU_LIBRARY=$PWD/$V/scratch_exp9.so timeout -k 10 200 python tools/scratch_two_streams.py --steps 30 --option scatter_cfg=$cfg 2>&1 | grep -v amdgpu.ids | grep -v "runs of unwritten" | cut -c1-220 | tail -4; } >> $OUT 2>&1
done
cat $OUT
