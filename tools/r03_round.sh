#!/bin/bash
# round-3 evidence run (GPU box): multi-GPU tests, --force-dist lines, kernel stats + PMC traffic + SQ counters
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r03h_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03h_rc.txt
for algo in phj cpra npj; do
  timeout -k 10 200 python bench.py --force-dist --algo $algo --steps 8 --warmup 2 --cpu-outer 0 > gpurun_out/r03_bench_force_dist_$algo.json 2> gpurun_out/r03_fd_$algo.err; echo "fd $algo rc=$?" >> gpurun_out/r03h_rc.txt
done
for n in 4 8; do timeout -k 10 200 python bench.py --force-dist --algo cpra --steps 8 --warmup 2 --cpu-outer 0 --exchange-slices $n > gpurun_out/r03_bench_force_dist_cpra_${n}slices.json 2>/dev/null; done
{ for o in "" "--transport loopback --world 2" "--unique" "--transport loopback --world 3 --slices 5"; do timeout -k 10 300 python tools/stress_cpra.py --steps 150 $o 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -4; done; } > gpurun_out/r03_stress_cpra.txt 2>&1
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1; echo "profile rc=$?" >> gpurun_out/r03h_rc.txt
python3 tools/collect_traffic.py --materialized > gpurun_out/r03_materialized_traffic.log 2>&1 && cp gpurun_out/traffic.json gpurun_out/r03_materialized_traffic.json
bash tools/pmc_sq.sh r03 > gpurun_out/r03_pmc_sq.log 2>&1; echo "pmc rc=$?" >> gpurun_out/r03h_rc.txt
cat gpurun_out/r03h_rc.txt
