#!/bin/bash
# Round 6: does the store loss of rounds 3-5 still show on today's boxes?  The platform's versions, then the slice pipeline with the
# forensic library variants (tools/build_variant.py): plain_priv (plain K6 stores + a private word in pass 1: round 4 lost stores in
# 39 of 40 steps at one priority) and plain (round 5: 1.5 x 10^-4 per step, 5 in 25 000 at one priority).
# usage (GPU box): bash tools/r06_rate.sh [steps_priv=200] [steps_plain=15000]
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=gpurun_out/r06_rate.txt
mkdir -p gpurun_out
V=hash_join_codes_knl_amd/lib/variants
quiet() { grep --line-buffered -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl"; }
{
echo "# store-loss rate on this box, $(date -u +%FT%RZ)"
echo "## platform"
uname -r
cat /sys/module/amdgpu/version 2>/dev/null || echo "amdgpu: in-tree module, no version file"
/opt/rocm/bin/rocm-smi --showfwinfo 2>/dev/null | grep -i "firmware\|GPU\[0\]" | head -40
/opt/rocm/bin/rocm-smi --showdriverversion 2>/dev/null | grep -i driver
python3 -c "import torch; print('torch', torch.__version__, 'hip', torch.version.hip)"
echo "## plain_priv, one priority for all streams"
HJGPU_DEBUG_FLAT_PRIORITIES=1 HJGPU_LIBRARY=$V/plain_priv.so timeout -k 10 300 python3 tools/stress_cpra.py --steps ${1:-200} --slices 8 2>&1 | quiet
echo "## plain_priv, shipped priorities, copying exchange"
HJGPU_LIBRARY=$V/plain_priv.so timeout -k 10 300 python3 tools/stress_cpra.py --steps ${1:-200} --slices 8 --option exchange_in_place=0 2>&1 | quiet
echo "## plain, one priority for all streams"
HJGPU_DEBUG_FLAT_PRIORITIES=1 HJGPU_LIBRARY=$V/plain.so timeout -k 10 600 python3 tools/stress_cpra.py --steps ${2:-15000} --slices 8 2>&1 | quiet
} > $out 2>&1
tail -60 $out
