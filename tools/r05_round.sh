#!/bin/bash
# Round-5 evidence run (GPU box), in the parts a 20-minute GPU call holds: tools/r05_round.sh tests | validation | validation2 | validation3 | profiles | dist | report
# Every artefact carries the kernel hash of the library that produced it; a part stops before it measures anything when the built
# library is not the tree's, and tools/r05_collect.sh refuses artefacts of another hash AND any sweep that contains a wrong step.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
hash=$(python3 - <<'PY'
import sys
import hash_join_codes_knl_amd as H
from hash_join_codes_knl_amd import build
lib, tree = H.kernel_hash(), build.kernel_hash()
if lib != tree:
    sys.stderr.write("library %s was not built from this tree (%s)\n" % (lib, tree))
    sys.exit(1)
print(tree)
PY
) || { echo "r05_round: refusing to collect evidence with a stale library"; exit 1; }
part=${1:-tests}
# one rc file per part: every gpurun call starts with an empty gpurun_out/ and the merge replaces files of the same name
rc=$out/r05_rc_$part.txt
echo "kernel hash $hash part $part $(date -u +%FT%RZ)" | tee $rc
note() { echo "$*" | tee -a $rc; }
quiet() { grep --line-buffered -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl"; }
stress() {   # every step of the slice pipeline checked; the tool exits 1 on a wrong step
  echo "## tools/stress_cpra.py $*" >> $out/$sweep
  timeout -k 10 600 python3 tools/stress_cpra.py "$@" 2>&1 | quiet | tee -a $out/$sweep | grep --line-buffered "^\.\.\."
  [ ${PIPESTATUS[0]} = 0 ] || { note "WRONG or failed: stress_cpra.py $*"; bad=1; }
}
case $part in
tests)
  PYTHONUNBUFFERED=1 timeout -k 10 1100 python3 -u -m pytest tests -m gpu -q -x > $out/r05_pytest.log 2>&1; note "pytest rc=$? $(tail -1 $out/r05_pytest.log)"
  ;;
validation)
  sweep=r05_validation.txt; bad=0
  { echo "# validation sweep, kernel hash $hash, $(date -u +%FT%RZ): random joins against numpy / the oracle, then EVERY step of the slice pipeline checked"
    echo "## HJ_FUZZ_SEED=9501 HJ_FUZZ_CASES=600 tests/test_gpu_fuzz.py"; } > $out/$sweep
  HJ_FUZZ_SEED=9501 HJ_FUZZ_CASES=600 timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -1 | tee -a $out/$sweep | grep -q " passed" || { note "fuzz failed"; bad=1; }
  echo "## HJ_FUZZ_SEED=9502 HJ_FUZZ_CASES=400 tests/test_gpu_multi.py -k random_multi" >> $out/$sweep
  HJ_FUZZ_SEED=9502 HJ_FUZZ_CASES=400 timeout -k 10 400 python3 -m pytest tests/test_gpu_multi.py -m gpu -q -k random_multi 2>&1 | tail -1 | tee -a $out/$sweep | grep -q " passed" || { note "random_multi failed"; bad=1; }
  stress --steps 15000 --slices 8
  stress --steps 15000 --slices 8 --unique
  note "validation bad=$bad"; [ $bad = 0 ] || exit 1
  ;;
validation2)
  sweep=r05_validation2.txt; bad=0
  echo "# validation sweep, second part, kernel hash $hash, $(date -u +%FT%RZ)" > $out/$sweep
  stress --steps 4000 --slices 8 --world 2 --transport loopback --unique
  stress --steps 2000 --slices 4 --world 8 --transport loopback
  stress --steps 5000 --slices 8 --option exchange_in_place=0
  stress --steps 5000 --slices 1
  HJGPU_DEBUG_FLAT_PRIORITIES=1 stress --steps 8000 --slices 8
  # the solo form (blocking hjgpu_phj on one stream with option solo: partial-line stores plain, DESIGN section 3 "Round 5"): every step checked
  echo "## tools/stress_single.py --algo phj --steps 10000 --solo" >> $out/$sweep
  timeout -k 10 400 python3 tools/stress_single.py --algo phj --steps 10000 --solo 2>&1 | quiet | tee -a $out/$sweep | grep --line-buffered "^\.\.\."
  [ ${PIPESTATUS[0]} = 0 ] || { note "WRONG or failed: stress_single.py"; bad=1; }
  note "validation2 bad=$bad"; [ $bad = 0 ] || exit 1
  ;;
validation3)
  # more of the same on the final library (the first two parts were cut to fit the round's GPU budget)
  sweep=r05_validation3.txt; bad=0
  echo "# validation sweep, third part, kernel hash $hash, $(date -u +%FT%RZ)" > $out/$sweep
  stress --steps 13000 --slices 8
  stress --steps 13000 --slices 8 --unique
  note "validation3 bad=$bad"; [ $bad = 0 ] || exit 1
  ;;
profiles)
  # kernel stats, PMC traffic of every leg of the N = 1 line, the default bench line with the traffic attached, SQ counters
  bash tools/profile_round.sh r05 > $out/r05_profile_round.log 2>&1; note "profile rc=$?"
  python3 tools/collect_traffic.py --materialized > $out/r05_materialized_traffic.log 2>&1 && cp $out/traffic.json $out/r05_materialized_traffic.json && cp $out/traffic.json profiles/r05_materialized_traffic.json
  python3 tools/collect_traffic.py --option unique=1 > $out/r05_unique_traffic.log 2>&1 && cp $out/traffic.json $out/r05_unique_traffic.json && cp $out/traffic.json profiles/r05_unique_traffic.json
  python3 bench.py --steps 20 --warmup 5 > $out/r05_bench.json 2> $out/r05_bench.err; note "bench rc=$?"
  bash tools/pmc_sq.sh r05 > $out/r05_pmc_sq.log 2>&1; note "pmc rc=$?"
  ;;
dist)
  # the multi-GPU entry points through RCCL at world 1 (bench.py --force-dist; the PHJ line carries secondary.cpra_multi =
  # BASELINE configs[4]'s per-rank shape), two processes on this one GPU (--rehearse-solo), all quoted workload shapes
  timeout -k 10 400 python3 bench.py --force-dist --steps 8 --warmup 2 --cpu-outer 0 > $out/r05_bench_force_dist_configs4.json 2> $out/r05_fd_phj.err; note "fd phj+configs4 rc=$?"
  for algo in cpra npj; do
    timeout -k 10 300 python3 bench.py --force-dist --algo $algo --steps 8 --warmup 2 --cpu-outer 0 > $out/r05_bench_force_dist_$algo.json 2> $out/r05_fd_$algo.err; note "fd $algo rc=$?"
  done
  timeout -k 10 300 python3 bench.py --force-dist --algo cpra --steps 8 --warmup 2 --cpu-outer 0 --exchange-slices 8 > $out/r05_bench_force_dist_cpra_8slices.json 2>/dev/null
  # a rank's share beyond two passes' reach, through the multi-GPU entry points at RCCL world 1: PHJ 1 G x 4 G (the rank's local join groups),
  # CPRA 700 M x 4 G on the grouped road (comm option cpra_grouped=2: wherever the planning rule groups) and on the one-level plan
  # (cpra_grouped=0; the default, 1, chooses it here: the road's extra pass does not pay at 3 fills per partition)
  timeout -k 10 300 python3 bench.py --force-dist --inner 1000000000 --outer 4000000000 --steps 3 --warmup 1 --cpu-outer 0 --no-secondary > $out/r05_bench_force_dist_phj_1G_4G.json 2> $out/r05_fd_phj_big.err; note "fd phj 1Gx4G rc=$?"
  timeout -k 10 300 python3 bench.py --force-dist --algo cpra --inner 700000000 --outer 4000000000 --steps 3 --warmup 1 --cpu-outer 0 --no-secondary --comm-option cpra_grouped=2 > $out/r05_bench_force_dist_cpra_700M_4G.json 2> $out/r05_fd_cpra_big.err; note "fd cpra 700Mx4G grouped rc=$?"
  timeout -k 10 300 python3 bench.py --force-dist --algo cpra --inner 700000000 --outer 4000000000 --steps 3 --warmup 1 --cpu-outer 0 --no-secondary --comm-option cpra_grouped=0 > $out/r05_bench_force_dist_cpra_700M_4G_ungrouped.json 2> $out/r05_fd_cpra_big0.err; note "fd cpra 700Mx4G ungrouped rc=$?"
  timeout -k 10 500 python3 bench.py --gpus 2 --rehearse-solo --steps 4 --warmup 1 --cpu-outer 0 --configs4-steps 2 > $out/r05_bench_rehearse_solo.json 2> $out/r05_rehearse.err; note "rehearse-solo rc=$?"
  ;;
report)
  python3 tools/report.py > $out/r05_report.md 2> $out/r05_report.err; note "report rc=$?"
  ;;
*) echo "unknown part $part"; exit 2;;
esac
tail -5 $rc
