#!/bin/bash
# Round 6, review item 1a: plain K6 stores (variant `plain`: tools/build_variant.py plain -DHJ_K6_STORE=0) in the slice pipeline with every
# stage audited and every slice's records read on the host as soon as the slice is done (communicator option debug_forensics = 2):
# a stage that lost tuples is looked at again AT ONCE, device quiet - by a fresh kernel and from a hipMemcpy on the host
# (hjgpu_audit_recheck) - while its buffers are still intact.  Memory wrong = lost stores; memory right = a stale read.
# usage (GPU box): bash tools/r06_lost_or_stale.sh [steps=25000] [seconds=800] [slices=8] [variant=plain]
#   variants (tools/build_variant.py): plain = -DHJ_K6_STORE=0 (every K6 store plain); plain16 = -DHJ_K6_STORE=4 (only K6's 16-byte whole-line
#   stores plain, its 8-byte partial-line stores non-temporal)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=gpurun_out/r06_lost_or_stale.txt
mkdir -p gpurun_out
quiet() { grep --line-buffered -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl"; }
{
echo "# lost store or stale read, $(date -u +%FT%RZ), $(uname -r)"
# which physical GPU this is (the rate of wrong steps differs from box to box by more than chance allows)
/opt/rocm/bin/rocm-smi --showuniqueid --showserial --showbus 2>/dev/null | grep -i "unique\|serial\|bus" | head -6
echo "## control: the product library, 30 steps"
timeout -k 10 200 python3 tools/stress_cpra.py --steps 30 --slices ${3:-8} --freeze 2>&1 | quiet
echo "## variant ${4:-plain} (tools/build_variant.py: plain = every K6 store plain, plain16 = only the 16-byte whole-line stores plain), one priority for all streams"
HJGPU_DEBUG_FLAT_PRIORITIES=1 HJGPU_LIBRARY=hash_join_codes_knl_amd/lib/variants/${4:-plain}.so timeout -k 10 ${2:-800} python3 tools/stress_cpra.py --steps ${1:-25000} --slices ${3:-8} --freeze 2>&1 | quiet
} > $out 2>&1
tail -50 $out
