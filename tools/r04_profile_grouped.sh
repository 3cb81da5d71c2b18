#!/bin/bash
# rocprofv3 kernel stats of a grouped plan (PHJ 1 G x 4 G: pass 0, then 16 two-pass joins)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_r04_grouped -- python3 bench.py --inner 1000000000 --outer 4000000000 --steps 3 --warmup 1 --no-secondary --cpu-outer 0 > $out/r04_grouped_prof_run.log 2>&1
echo "rocprof rc=$?"
f=$(find $out/prof_r04_grouped -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; exit 1; }
cp "$f" $out/r04_phj_1G_4G_grouped_kernel_stats.csv
head -8 "$f" | cut -c1-200
