#!/bin/bash
# Soak of the round's new grouped paths on the final library (every step checked; exits 1 on a wrong step):
#   hjgpu_cpra_multi on the grouped road (comm option cpra_grouped=2, 8 groups per rank) at RCCL world 1 and loopback world 2,
#   hjgpu_phj_async / hjgpu_cpra_async with a grouped plan (the context's worker thread), hjgpu_phj blocking with a grouped plan.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=gpurun_out/r05_grouped_soak.txt
hash=$(python3 -c "import hash_join_codes_knl_amd as H; print(H.kernel_hash())")
echo "# grouped paths soak, kernel hash $hash, $(date -u +%FT%RZ)" > $out
bad=0
quiet() { grep --line-buffered -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl"; }
G="--ctx-option group_from=1000 --ctx-option group_always=1 --ctx-option group_inner=8000000"
run() { echo "## $*" >> $out; timeout -k 10 400 "$@" 2>&1 | quiet | tee -a $out | grep --line-buffered "^\.\.\.\|wrong"; [ ${PIPESTATUS[0]} = 0 ] || { echo "WRONG or failed: $*" | tee -a $out; bad=1; }; }
run python3 tools/stress_cpra.py --steps 2000 --slices 8 --option cpra_grouped=2 $G
run python3 tools/stress_cpra.py --steps 1000 --slices 4 --world 2 --transport loopback --option cpra_grouped=2 $G
run python3 tools/stress_single.py --algo phj --steps 2000 --enqueue-only $G
run python3 tools/stress_single.py --algo cpra --steps 1000 --enqueue-only $G
run python3 tools/stress_single.py --algo phj --steps 1000 $G
echo "soak bad=$bad" | tee -a $out
exit $bad
