cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_grouped.py -x -q > gpurun_out/r04_grouped_tests.txt 2>&1; echo "grouped tests rc=$?"; tail -5 gpurun_out/r04_grouped_tests.txt
timeout -k 10 500 python tools/grouped_sweep.py 4000000000 1000000000 700000000 > gpurun_out/r04_grouped_sweep_4g.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/r04_grouped_sweep_4g.txt
timeout -k 10 500 python tools/grouped_sweep.py 2000000000 700000000 > gpurun_out/r04_grouped_sweep_2g.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/r04_grouped_sweep_2g.txt
