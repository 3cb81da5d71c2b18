cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_suite.txt 2>&1; echo "gpu suite rc=$?"; tail -4 gpurun_out/r04_gpu_suite.txt
timeout -k 10 200 python bench.py --inner 1000000000 --outer 4000000000 --steps 3 --warmup 1 --no-secondary --cpu-outer 0 > gpurun_out/r04_bench_1g_4g.json 2> gpurun_out/r04_bench_1g_4g.err; echo "bench 1g4g rc=$?"; cut -c1-1500 gpurun_out/r04_bench_1g_4g.json
