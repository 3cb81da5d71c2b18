#!/bin/bash
# SQ counters (instruction mix, LDS bank conflicts, busy cycles) of the PHJ kernels; run on the GPU box.
# usage: tools/pmc_sq.sh <tag>    -> gpurun_out/pmc_sq_<tag>.csv  (kernel,dispatch,counter,value)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$1
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${out}_$(echo $set | cut -c1-12 | tr ' ' _) -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-outer 0 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(float)
seen = {}
for f in glob.glob("${out}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if not any(s in name for s in ("scatter", "join_kernel", "hist2")):
            continue
        key = (name, r["Counter_Name"])
        seen.setdefault(key, r["Dispatch_Id"])
        if r["Dispatch_Id"] == seen[key] or True:
            rows[(name, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
with open("${out}.csv", "w", newline="") as o:
    w = csv.writer(o)                                # kernel names contain commas
    w.writerow(["kernel", "dispatch", "counter", "value"])
    for (k, d, c), v in sorted(rows.items()):
        w.writerow([k, d, c, "%.0f" % v])
# one line per kernel: the largest dispatch of each
best = {}
for (k, d, c), v in rows.items():
    if c == "SQ_BUSY_CYCLES" and v > best.get(k, (0, None))[0]:
        best[k] = (v, d)
PY
echo written ${out}.csv
