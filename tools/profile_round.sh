#!/bin/bash
# One round's profiling artefacts; run on the GPU box: tools/profile_round.sh <tag>   (e.g. r03)
# Writes gpurun_out/<tag>_*: rocprofv3 --kernel-trace --stats summaries of the PHJ headline run, NPJ, one-GPU CPRA and
# the materialising PHJ, and the PMC traffic files (separate FETCH_SIZE / WRITE_SIZE passes, tools/collect_traffic.py).
# Copy what is to be judged into profiles/.
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
[ -z "$root" ] && root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {   # name, program args...
  name=$1; shift
  rm -rf $out/prof_${tag}_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_$name -- "$@" > $out/${tag}_${name}_run.log 2>&1 || return 1
  f=$(find $out/prof_${tag}_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $out/${tag}_${name}_kernel_stats.csv
  echo "$name: $(grep -c . $out/${tag}_${name}_kernel_stats.csv) rows"
}
stats phj_64M_1G python3 $root/bench.py --steps 10 --warmup 2 --cpu-outer 0 --no-secondary &&
stats npj_64M_1G python3 $root/bench.py --algo npj --steps 5 --warmup 1 --cpu-outer 0 &&
stats cpra_64M_1G python3 $root/bench.py --algo cpra --steps 5 --warmup 1 --cpu-outer 0 &&
stats materialized_64M_1G python3 $root/tools/run_materialized.py 5 &&
cd $root &&
python3 tools/collect_traffic.py && cp gpurun_out/traffic.json gpurun_out/${tag}_traffic.json &&
python3 tools/collect_traffic.py --algo npj && cp gpurun_out/traffic.json gpurun_out/${tag}_npj_traffic.json &&
python3 tools/collect_traffic.py --algo cpra && cp gpurun_out/traffic.json gpurun_out/${tag}_cpra_traffic.json &&
# the bench line attaches profiles/<round>_traffic.json (and refuses it when the kernel hash differs): this run's files
round=${tag:0:3} &&
cp gpurun_out/${tag}_traffic.json profiles/${round}_traffic.json &&
cp gpurun_out/${tag}_npj_traffic.json profiles/${round}_npj_traffic.json &&
cp gpurun_out/${tag}_cpra_traffic.json profiles/${round}_cpra_traffic.json &&
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "profile_round done rc=$?"
