#!/bin/bash
# after tools/r03_round.sh on the GPU box: copy what is to be judged from gpurun_out/ into profiles/ and regenerate the derived files
cd "$(dirname "$0")/.."
for f in r03_bench.json r03_bench_force_dist_phj.json r03_bench_force_dist_cpra.json r03_bench_force_dist_cpra_4slices.json r03_bench_force_dist_cpra_8slices.json r03_bench_force_dist_npj.json r03_cpra_64M_1G_kernel_stats.csv r03_materialized_64M_1G_kernel_stats.csv r03_npj_64M_1G_kernel_stats.csv r03_phj_64M_1G_kernel_stats.csv r03_traffic.json r03_npj_traffic.json r03_cpra_traffic.json r03_materialized_traffic.json r03_stress_cpra.txt; do cp gpurun_out/$f profiles/$f; done
cp gpurun_out/pmc_sq_r03.csv profiles/r03_pmc_sq.csv
python tools/pmc_sq_summary.py profiles/r03_pmc_sq.csv > profiles/r03_pmc_sq_summary.txt
{ echo "# tools/kernel_resources.py (hipcc -Rpass-analysis=kernel-resource-usage, gfx950) on the round-3 sources: VGPRs, spills, scratch bytes per lane, waves per SIMD"; for f in partition_kernels join_kernels npj_kernels gen_kernels; do echo "## $f.hip"; python tools/kernel_resources.py hash_join_codes_knl_amd/csrc/$f.hip 2>&1; done; } > profiles/r03_kernel_resources.txt
tail -2 gpurun_out/r03h_pytest.log
