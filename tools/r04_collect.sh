#!/bin/bash
# After tools/r04_round.sh on the GPU box: copy what is to be judged from gpurun_out/ into profiles/ - but only artefacts
# produced by THIS tree's kernels (round 3 committed a validation sweep of an earlier library under the final commit's
# title).  JSON artefacts carry "kernel_hash"; text artefacts are covered by gpurun_out/r04_rc.txt ("kernel hash ...",
# written by the run before anything else, after it compared the library with the tree).
cd "$(dirname "$0")/.."
tree=$(python3 -c "from hash_join_codes_knl_amd import build; print(build.kernel_hash())")
run=$(sed -n 's/^kernel hash //p' gpurun_out/r04_rc.txt | head -1)
if [ "$run" != "$tree" ]; then echo "REFUSED: the evidence run used kernels $run, the tree is $tree"; exit 1; fi
ok=1
for f in r04_bench.json r04_bench_force_dist_configs4.json r04_bench_force_dist_cpra.json r04_bench_force_dist_cpra_4slices.json r04_bench_force_dist_cpra_8slices.json r04_bench_force_dist_npj.json r04_bench_rehearse_solo.json r04_traffic.json r04_npj_traffic.json r04_cpra_traffic.json r04_materialized_traffic.json; do
  [ -s gpurun_out/$f ] || { echo "missing: $f"; ok=0; continue; }
  h=$(python3 -c "import json,sys; print(json.load(open('gpurun_out/$f')).get('kernel_hash'))")
  if [ "$h" != "$tree" ]; then echo "REFUSED: $f carries kernel hash $h, the tree is $tree"; ok=0; continue; fi
  cp gpurun_out/$f profiles/$f
done
for f in r04_cpra_64M_1G_kernel_stats.csv r04_materialized_64M_1G_kernel_stats.csv r04_npj_64M_1G_kernel_stats.csv r04_phj_64M_1G_kernel_stats.csv r04_validation.txt r04_npj_build_ceiling.txt r04_report.md r04_grouped_sweep.txt; do
  [ -s gpurun_out/$f ] || { echo "missing: $f"; ok=0; continue; }
  cp gpurun_out/$f profiles/$f
done
grep -q "$tree" profiles/r04_grouped_sweep.txt || { echo "REFUSED: r04_grouped_sweep.txt does not name kernel hash $tree"; ok=0; }
# (measured earlier in the round and removed from the library: appended again after every collection)
cat >> profiles/r04_npj_build_ceiling.txt <<'TXT'

# measured earlier in the round and removed (kernel hash 71b4fd47715f8ac9): a K2 that issues its claims WITHOUT looking first, 1 / 2 / 4 / 8 in flight per lane
# (with the line-hashed table every key of a 64-byte line starts at the line's first bucket: every second blind CAS loses and walks on)
NPJ 64000000 x 1000000000, npj_build=1: build (table clear + claims) 4.980 ms, probe 20.228 ms, total 25.213 ms
NPJ 64000000 x 1000000000, npj_build=2: build (table clear + claims) 14.195 ms, probe 20.164 ms, total 34.366 ms
NPJ 64000000 x 1000000000, npj_build=4: build (table clear + claims) 37.463 ms, probe 20.201 ms, total 57.669 ms
NPJ 64000000 x 1000000000, npj_build=8: build (table clear + claims) 37.377 ms, probe 20.269 ms, total 57.653 ms
TXT
grep -q "$tree" profiles/r04_validation.txt || { echo "REFUSED: r04_validation.txt does not name kernel hash $tree"; rm -f profiles/r04_validation.txt; ok=0; }
cp gpurun_out/pmc_sq_r04.csv profiles/r04_pmc_sq.csv && python tools/pmc_sq_summary.py profiles/r04_pmc_sq.csv > profiles/r04_pmc_sq_summary.txt
{ echo "# tools/kernel_resources.py (hipcc -Rpass-analysis=kernel-resource-usage, gfx950) on the round-4 sources (kernel hash $tree): VGPRs, spills, scratch bytes per lane, waves per SIMD"; for f in partition_kernels join_kernels npj_kernels gen_kernels; do echo "## $f.hip"; python tools/kernel_resources.py hash_join_codes_knl_amd/csrc/$f.hip 2>&1; done; } > profiles/r04_kernel_resources.txt
tail -3 gpurun_out/r04_pytest.log
[ $ok = 1 ] && echo "collected for kernel hash $tree" || echo "collected WITH GAPS for kernel hash $tree"
