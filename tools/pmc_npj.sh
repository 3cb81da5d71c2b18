#!/bin/bash
# PMC counters of the NPJ probe kernels (run on the GPU box): L2 requests, memory-side read requests
# usage: tools/pmc_npj.sh <tag> [env assignments are inherited]
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/npj_pmc_$1
rocprofv3 --pmc TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --algo npj --steps 1 --warmup 1 --cpu-outer 0 > $out.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob,collections
per=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "npj" in r["Kernel_Name"]:
            per[r["Kernel_Name"][:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in per.items():
    print("$1", k, {c:"%.3fG" % (sum(x)/len(x)/1e9) for c,x in v.items()})
PY
