#!/bin/bash
# Round 4, review item 5: counters of a FAST and a SLOW 8.5 GB block of one process (tools/ubench_placement_counters.hip).
# One --pmc group per pass, the program directly after "--".  -> gpurun_out/r04_placement_counters.txt
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
exe=$root/hash_join_codes_knl_amd/lib/ubench_placement_counters
cd /tmp && export TMPDIR=/tmp
{
echo "# r04 placement counters, $(date -u +%Y-%m-%dT%H:%MZ): fill_fast_kernel / fill_slow_kernel = the same streaming fill of 8.5 GB into the fastest / slowest of 8 blocks held side by side"
echo "## timing run (no profiler)"
timeout -k 10 120 $exe 8 5
} > $out/r04_placement_counters.txt 2>&1
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_REQ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_BUSY_sum" \
           "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITEBACK_sum TCC_CYCLE_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCC_EA0_WRREQ_DRAM_sum TCC_STREAMING_REQ_sum"; do
  i=$((i + 1))
  rm -rf $out/plc_$i
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/plc_$i -- $exe 8 5 0 > $out/plc_$i.log 2>&1
  echo "pass $i rc=$? ($set)" >> $out/r04_placement_counters.txt
done
cd $root
python3 - >> $out/r04_placement_counters.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$out/plc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        kind = "fast" if "fill_fast" in k else "slow" if "fill_slow" in k else None
        if kind:
            a = acc[(r["Counter_Name"], kind)]
            a[0] += float(r["Counter_Value"]); a[1] += 1
print("## counters per launch (mean over the launches seen), the same fill into the FAST and into the SLOW block")
print("%-46s %16s %16s %8s" % ("counter", "fast", "slow", "slow/fast"))
names = sorted({c for c, _ in acc})
for c in names:
    f, s = acc[(c, "fast")], acc[(c, "slow")]
    if not f[1] or not s[1]: continue
    fv, sv = f[0] / f[1], s[0] / s[1]
    print("%-46s %16.0f %16.0f %8.3f" % (c, fv, sv, sv / fv if fv else float("nan")))
PY
tail -45 $out/r04_placement_counters.txt
