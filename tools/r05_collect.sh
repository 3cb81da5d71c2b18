#!/bin/bash
# After the parts of tools/r05_round.sh: copy what is to be judged from gpurun_out/ into profiles/ - only artefacts produced by
# THIS tree's kernels, and never silently a sweep with a wrong step: a validation file that contains WRONG makes this script
# exit 1 (round 4's collection committed one without a word).
cd "$(dirname "$0")/.."
tree=$(python3 -c "from hash_join_codes_knl_amd import build; print(build.kernel_hash())")
ok=1
for f in r05_validation.txt r05_validation2.txt; do
  [ -s gpurun_out/$f ] || { echo "missing: $f"; ok=0; continue; }
  grep -q "kernel hash $tree" gpurun_out/$f || { echo "REFUSED: $f does not name kernel hash $tree"; ok=0; continue; }
  if grep -q "WRONG" gpurun_out/$f; then echo "WRONG STEPS in $f (copied, and this script fails):"; grep "WRONG\|steps wrong" gpurun_out/$f; ok=0; fi
  grep -v "^\.\.\." gpurun_out/$f > profiles/$f
done
if [ -s gpurun_out/r05_validation3.txt ]; then      # optional third part
  f=r05_validation3.txt
  if ! grep -q "kernel hash $tree" gpurun_out/$f; then echo "REFUSED: $f does not name kernel hash $tree"; ok=0
  else
    if grep -q "WRONG" gpurun_out/$f; then echo "WRONG STEPS in $f (copied, and this script fails):"; grep "WRONG\|steps wrong" gpurun_out/$f; ok=0; fi
    grep -v "^\.\.\." gpurun_out/$f > profiles/$f
  fi
fi
for f in r05_bench.json r05_bench_force_dist_configs4.json r05_bench_force_dist_cpra.json r05_bench_force_dist_npj.json r05_bench_force_dist_cpra_8slices.json r05_bench_rehearse_solo.json \
         r05_bench_force_dist_phj_1G_4G.json r05_bench_force_dist_cpra_700M_4G.json r05_bench_force_dist_cpra_700M_4G_ungrouped.json \
         r05_traffic.json r05_npj_traffic.json r05_cpra_traffic.json r05_materialized_traffic.json r05_unique_traffic.json; do
  [ -s gpurun_out/$f ] || { echo "missing: $f"; ok=0; continue; }
  h=$(python3 -c "import json,sys; print(json.load(open('gpurun_out/$f')).get('kernel_hash'))")
  if [ "$h" != "$tree" ]; then echo "REFUSED: $f carries kernel hash $h, the tree is $tree"; ok=0; continue; fi
  cp gpurun_out/$f profiles/$f
done
# (the parts' own rc files, or - for parts run before the round script wrote one per part - the call's log)
cat gpurun_out/r05_rc_profiles.txt gpurun_out/r05_profiles_call.log 2>/dev/null | grep -q "kernel hash $tree part profiles" || { echo "REFUSED: the profiles part did not run on kernel hash $tree"; ok=0; }
for f in r05_cpra_64M_1G_kernel_stats.csv r05_materialized_64M_1G_kernel_stats.csv r05_npj_64M_1G_kernel_stats.csv r05_phj_64M_1G_kernel_stats.csv r05_report.md; do
  [ -s gpurun_out/$f ] || { echo "missing: $f"; ok=0; continue; }
  cp gpurun_out/$f profiles/$f
done
[ -s gpurun_out/pmc_sq_r05.csv ] && cp gpurun_out/pmc_sq_r05.csv profiles/r05_pmc_sq.csv && python3 tools/pmc_sq_summary.py profiles/r05_pmc_sq.csv > profiles/r05_pmc_sq_summary.txt
{ echo "# tools/kernel_resources.py (hipcc -Rpass-analysis=kernel-resource-usage, gfx950) on the round-5 sources (kernel hash $tree): VGPRs, spills, scratch bytes per lane, waves per SIMD"
  for f in partition_kernels.hip join_kernels.hip npj_kernels.hip audit_kernels.hip; do python3 tools/kernel_resources.py hash_join_codes_knl_amd/csrc/$f; done; } > profiles/r05_kernel_resources.txt
for f in $(grep -l "^kernel hash $tree" gpurun_out/r05_rc*.txt gpurun_out/r05_*_call.log 2>/dev/null); do
  awk -v t="$tree" '/^kernel hash/ { on = ($3 == t) } on' $f | grep "^kernel hash\|^[a-z0-9+ -]*rc=\|bad="
done | sort -u
[ $ok = 1 ] && echo "collected for kernel hash $tree" || { echo "collected WITH GAPS OR WRONG STEPS for kernel hash $tree"; exit 1; }
