#!/bin/bash
# Round-6 evidence run (GPU box), in parts a 20-minute GPU call holds:  tools/r06_round.sh tests | profiles | big | dist
# (the soaks: tools/r06_soak.sh).  A part stops before it measures anything when the built library is not the tree's
# (hjgpu_library_hash() != hash_join_codes_knl_amd.build.library_hash()); every artefact names the library hash, and
# tools/r06_collect.sh refuses artefacts of another one and any log that contains a wrong step.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp NCCL_SOCKET_IFNAME=lo
out=gpurun_out
mkdir -p $out
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
) || { echo "r06_round: refusing to collect evidence with a stale library"; exit 1; }
part=${1:-tests}
rc=$out/r06_rc_$part.txt
echo "library hash $hash part $part $(date -u +%FT%RZ) $(uname -r)" | tee $rc
note() { echo "$*" | tee -a $rc; }
case $part in
tests)
  PYTHONUNBUFFERED=1 timeout -k 10 1100 python3 -u -m pytest tests -m gpu -q -x > $out/r06_pytest.log 2>&1; note "pytest rc=$? $(tail -1 $out/r06_pytest.log)"
  ;;
profiles)
  # kernel stats, PMC traffic of every leg of the N = 1 line, the default bench line with the traffic attached, SQ counters
  bash tools/profile_round.sh r06 > $out/r06_profile_round.log 2>&1; note "profile rc=$?"
  python3 tools/collect_traffic.py --materialized > $out/r06_materialized_traffic.log 2>&1 && cp $out/traffic.json $out/r06_materialized_traffic.json && cp $out/traffic.json profiles/r06_materialized_traffic.json
  python3 tools/collect_traffic.py --option unique=1 > $out/r06_unique_traffic.log 2>&1 && cp $out/traffic.json $out/r06_unique_traffic.json && cp $out/traffic.json profiles/r06_unique_traffic.json
  python3 bench.py --steps 20 --warmup 5 > $out/r06_bench.json 2> $out/r06_bench.err; note "bench rc=$?"
  bash tools/pmc_sq.sh r06 > $out/r06_pmc_sq.log 2>&1; note "pmc rc=$?"
  ;;
big)
  # grouped plans at full size, enqueue-only (planned on the device) and blocking; the host-planned form beside them
  timeout -k 10 300 python3 bench.py --inner 1000000000 --outer 4000000000 --steps 3 --warmup 1 --cpu-outer 0 --no-secondary --enqueue-only > $out/r06_bench_1G_4G_async.json 2> $out/r06_big1.err; note "1Gx4G enqueue-only rc=$?"
  timeout -k 10 300 python3 bench.py --inner 1000000000 --outer 4000000000 --steps 3 --warmup 1 --cpu-outer 0 --no-secondary > $out/r06_bench_1G_4G_blocking.json 2> $out/r06_big2.err; note "1Gx4G blocking rc=$?"
  timeout -k 10 300 python3 bench.py --inner 1000000000 --outer 4000000000 --steps 3 --warmup 1 --cpu-outer 0 --no-secondary --option group_device=0 > $out/r06_bench_1G_4G_host_planned.json 2> $out/r06_big3.err; note "1Gx4G host-planned rc=$?"
  ;;
dist)
  # the multi-GPU entry points through RCCL at world 1 (bench.py --force-dist), all quoted workload shapes, and a rank's share beyond two passes' reach
  timeout -k 10 400 python3 bench.py --force-dist --steps 8 --warmup 2 --cpu-outer 0 > $out/r06_bench_force_dist_configs4.json 2> $out/r06_fd_phj.err; note "fd phj+configs4 rc=$?"
  for algo in cpra npj; do
    timeout -k 10 300 python3 bench.py --force-dist --algo $algo --steps 8 --warmup 2 --cpu-outer 0 > $out/r06_bench_force_dist_$algo.json 2> $out/r06_fd_$algo.err; note "fd $algo rc=$?"
  done
  timeout -k 10 300 python3 bench.py --force-dist --inner 1000000000 --outer 4000000000 --steps 3 --warmup 1 --cpu-outer 0 --no-secondary > $out/r06_bench_force_dist_phj_1G_4G.json 2> $out/r06_fd_phj_big.err; note "fd phj 1Gx4G rc=$?"
  timeout -k 10 300 python3 bench.py --force-dist --algo cpra --inner 700000000 --outer 4000000000 --steps 3 --warmup 1 --cpu-outer 0 --no-secondary --comm-option cpra_grouped=2 > $out/r06_bench_force_dist_cpra_700M_4G.json 2> $out/r06_fd_cpra_big.err; note "fd cpra 700Mx4G grouped rc=$?"
  ;;
*) echo "unknown part $part"; exit 2;;
esac
tail -6 $rc
