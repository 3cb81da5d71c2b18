#!/bin/bash
# Round 6, review item 1a: is a "lost" K6 store wrong IN MEMORY (lost) or only in what the next kernel reads (stale)?
# The two-stream harness (no pipeline, no communicator) with forensic library variants (tools/build_variant.py):
#   plain_priv  plain K6 stores + one private word in pass 1 (round 4's amplifier: a third of the steps wrong)
#   nt_priv     the product's non-temporal stores + the same private word (does nt close the amplified form too?)
#   plain       plain K6 stores, no private word (round 5's 1.5 x 10^-4 per pipeline step)
# usage (GPU box): bash tools/r06_forensics.sh [steps_priv=60] [steps_plain=1500]
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=gpurun_out/r06_forensics.txt
mkdir -p gpurun_out
V=hash_join_codes_knl_amd/lib/variants
{
echo "# lost store or stale read?  $(date -u +%FT%RZ)"
echo "## plain_priv, default copy engines (hipMemcpy = SDMA)"
HJGPU_LIBRARY=$V/plain_priv.so timeout -k 10 300 python3 tools/scratch_two_streams.py --steps ${1:-60} --recheck --quiet-after 8 2>&1 | grep -v amdgpu.ids
echo "## plain_priv, HSA_ENABLE_SDMA=0 (hipMemcpy = a blit kernel, through the L2s)"
HSA_ENABLE_SDMA=0 HJGPU_LIBRARY=$V/plain_priv.so timeout -k 10 300 python3 tools/scratch_two_streams.py --steps 30 --recheck --quiet-after 3 2>&1 | grep -v amdgpu.ids
echo "## nt_priv: the product's stores with the same private word"
HJGPU_LIBRARY=$V/nt_priv.so timeout -k 10 300 python3 tools/scratch_two_streams.py --steps ${1:-60} --recheck 2>&1 | grep -v amdgpu.ids
echo "## plain: no private word"
HJGPU_LIBRARY=$V/plain.so timeout -k 10 600 python3 tools/scratch_two_streams.py --steps ${2:-1500} --recheck 2>&1 | grep -v amdgpu.ids
} > $out 2>&1
tail -40 $out
