#!/bin/bash
# Round 4: what does the cap on the placement search cost?  Pass-1 time (ms_scatter1 of PHJ 64 M x 1 G, best of 4 steps) and the
# wall clock of the workspace growth (ms_reserve) over FRESH PROCESSES, for `placement` = 1 / 4 / 8 / 12 candidates.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_placement_dist.txt
echo "# tools/r04_placement_dist.sh $(date -u +%FT%RZ), kernel hash $(python -c 'from hash_join_codes_knl_amd import build; print(build.kernel_hash())')" > $out
for i in 1 2 3 4 5 6 7 8 9 10; do
  for p in 1 4 8 12; do
    HJGPU_PLACEMENT=$p timeout -k 5 120 python tools/alloc_luck.py serial 1 phj 2>&1 | sed "s/^/placement=$p process $i: /" >> $out || exit 1
  done
done
python - <<'PY' >> gpurun_out/r04_placement_dist.txt
import re, collections
d = collections.defaultdict(list); r = collections.defaultdict(list)
for line in open("gpurun_out/r04_placement_dist.txt"):
    m = re.match(r"placement=(\d+) process \d+: .*scatter1 ([\d.]+) .*reserve ([\d.]+)", line)
    if m: d[int(m.group(1))].append(float(m.group(2))); r[int(m.group(1))].append(float(m.group(3)))
print("# placement: pass 1 ms (sorted) | reserve ms mean / max")
for p in sorted(d):
    print("# %2d: %s | %.0f / %.0f" % (p, " ".join("%.2f" % x for x in sorted(d[p])), sum(r[p]) / len(r[p]), max(r[p])))
PY
