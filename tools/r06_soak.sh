#!/bin/bash
# Round 6, review item 1d: >= 20 000 checked steps through each pipeline that runs several streams side by side, on the library of THIS
# tree (the part stops before it measures anything when the built library is another one).  Every log starts with the library hash.
#   tools/r06_soak.sh npj | phj | phj2 | host | grouped | cpra | cprarows | phjrows | phjgrouped | cpragrouped | blockgrouped      [steps]
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp NCCL_SOCKET_IFNAME=lo
mkdir -p gpurun_out
hash=$(python3 - <<'PY'
import sys
import hash_join_codes_knl_amd as H
from hash_join_codes_knl_amd import build
lib, tree = H.library_hash(), build.library_hash()
if lib != tree:
    sys.stderr.write("library %s was not built from this tree (%s)\n" % (lib, tree))
    sys.exit(1)
print(tree)
PY
) || { echo "r06_soak: refusing to collect evidence with a stale library"; exit 1; }
part=${1:-npj}; steps=${2:-20000}
out=gpurun_out/r06_soak_$part.txt
quiet() { grep --line-buffered -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl"; }
echo "# soak $part, library hash $hash, $steps steps, $(date -u +%FT%RZ), $(uname -r)" > $out
run() { echo "## $*" >> $out; timeout -k 10 ${T:-900} "$@" 2>&1 | quiet >> $out; rc=${PIPESTATUS[0]}; [ $rc = 0 ] || { echo "WRONG or failed (rc $rc): $*" >> $out; bad=1; }; }
bad=0
case $part in
npj)     run python3 tools/stress_cpra.py --algo npj --steps $steps ;;
phj)     run python3 tools/stress_cpra.py --algo phj --steps $steps ;;
phj2)    run python3 tools/stress_cpra.py --algo phj --steps $steps --world 2 --transport loopback ;;
cpra)    run python3 tools/stress_cpra.py --algo cpra --steps $steps --slices 8 ;;
cprarows) run python3 tools/stress_cpra.py --algo cpra --steps $steps --slices 8 --rows ;;
phjrows) run python3 tools/stress_cpra.py --algo phj --steps $steps --rows ;;
# grouped plans (planned on the device) as the local joins of the multi-GPU calls, and blocking on one stream
phjgrouped)  run python3 tools/stress_cpra.py --algo phj --steps $steps --ctx-option group_from=1000 --ctx-option group_always=1 --ctx-option group_inner=16000000 ;;
cpragrouped) run python3 tools/stress_cpra.py --algo cpra --steps $steps --slices 4 --option cpra_grouped=2 --ctx-option group_from=1000 --ctx-option group_always=1 --ctx-option group_inner=16000000 ;;
blockgrouped) run python3 tools/stress_single.py --algo phj --steps $steps --ctx-option group_from=1000 --ctx-option group_always=1 --ctx-option group_inner=16000000 ;;
host)    run python3 tools/stress_host_rows.py --algo phj --steps $steps ;;
grouped) run python3 tools/stress_async_grouped.py --steps $((steps / 2)) --depth 2 ;;
*) echo "unknown part $part"; exit 2;;
esac
echo "soak $part bad=$bad" >> $out
tail -6 $out
[ $bad = 0 ]
