#!/bin/bash
# Round-4 evidence run (GPU box), ONE pass at the end of the round.  Every artefact carries the kernel hash of the library
# that produced it; the run stops before anything is measured when the built library is not the tree's
# (hjgpu_kernel_hash() != hash_join_codes_knl_amd.build.kernel_hash()), and tools/r04_collect.sh refuses to copy an
# artefact whose hash differs from the tree's.
cd $GRAFT_REPO_ROOT
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
) || { echo "r04_round: refusing to collect evidence with a stale library"; exit 1; }
echo "kernel hash $hash" | tee $out/r04_rc.txt
note() { echo "$*" >> $out/r04_rc.txt; }
# 1. the gpu suite
python -m pytest tests -m gpu -q > $out/r04_pytest.log 2>&1; note "pytest rc=$?"
# 2. validation sweep on THIS library
{ echo "# validation sweep, kernel hash $hash: HJ_FUZZ_SEED=9101 HJ_FUZZ_CASES=600 tests/test_gpu_fuzz.py; HJ_FUZZ_SEED=9102 HJ_FUZZ_CASES=400"
  echo "# tests/test_gpu_multi.py -k random_multi; tools/stress_cpra.py (every step of the slice pipeline checked)"
  HJ_FUZZ_SEED=9101 HJ_FUZZ_CASES=600 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -1
  HJ_FUZZ_SEED=9102 HJ_FUZZ_CASES=400 timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -m gpu -q -k random_multi 2>&1 | tail -1
  for o in "--steps 1000" "--steps 600 --transport loopback --world 8 --slices 4" "--steps 600 --unique" "--steps 600 --option exchange_in_place=0" "--steps 400 --transport loopback --world 2"; do
    timeout -k 10 400 python tools/stress_cpra.py $o 2>&1 | grep "steps wrong\|WRONG\|kernel hash"
  done; } > $out/r04_validation.txt 2>&1
note "validation done"
# 3. the multi-GPU entry points through RCCL at world 1 (bench.py --force-dist); the PHJ line carries secondary.cpra_multi
#    (BASELINE configs[4]'s per-rank shape); --rehearse-solo: the N > 1 control flow with two processes on this one GPU
timeout -k 10 400 python bench.py --force-dist --steps 8 --warmup 2 --cpu-outer 0 > $out/r04_bench_force_dist_configs4.json 2> $out/r04_fd_phj.err; note "fd phj+configs4 rc=$?"
for algo in cpra npj; do
  timeout -k 10 300 python bench.py --force-dist --algo $algo --steps 8 --warmup 2 --cpu-outer 0 > $out/r04_bench_force_dist_$algo.json 2> $out/r04_fd_$algo.err; note "fd $algo rc=$?"
done
for n in 4 8; do timeout -k 10 300 python bench.py --force-dist --algo cpra --steps 8 --warmup 2 --cpu-outer 0 --exchange-slices $n > $out/r04_bench_force_dist_cpra_${n}slices.json 2>/dev/null; done
timeout -k 10 600 python bench.py --gpus 2 --rehearse-solo --steps 4 --warmup 1 --cpu-outer 0 --configs4-steps 2 > $out/r04_bench_rehearse_solo.json 2> $out/r04_rehearse.err; note "rehearse-solo rc=$?"
# 4. kernel stats, PMC traffic, the default bench line with the traffic attached
bash tools/profile_round.sh r04 > $out/r04_profile_round.log 2>&1; note "profile rc=$?"
python3 tools/collect_traffic.py --materialized > $out/r04_materialized_traffic.log 2>&1 && cp $out/traffic.json $out/r04_materialized_traffic.json
bash tools/pmc_sq.sh r04 > $out/r04_pmc_sq.log 2>&1; note "pmc rc=$?"
# 5. the NPJ build's ceiling, all quoted workload shapes
python tools/npj_build_ceiling.py 2>&1 | grep -v amdgpu.ids > $out/r04_npj_build_ceiling.txt
python tools/report.py > $out/r04_report.md 2> $out/r04_report.err; note "report rc=$?"
# 6. grouped plans against two passes with multi-fill tables (the sizes the cost estimate in grouped_groups() was fitted on)
{ timeout -k 10 300 python tools/grouped_sweep.py 1000000000 512000000 1000000000; timeout -k 10 300 python tools/grouped_sweep.py 4000000000 700000000 1000000000; } 2>&1 | grep -v amdgpu.ids > $out/r04_grouped_sweep.txt; note "grouped sweep done"
cat $out/r04_rc.txt
