#!/usr/bin/env python3
"""K6 pass 1 takes 2.9 or 3.3-3.4 ms on the same data depending on WHICH allocation its output twin lives in (the
virtual offset inside an allocation does not matter: profiles/r02_shift_sweep.txt).  This looks at the distribution:
contexts created one after the other (each frees its workspace before the next is made: `serial`) or kept alive
(`stacked`: every new workspace lands somewhere else).  usage: python tools/alloc_luck.py [serial|stacked] [n] [phj|npj|cpra]; HJGPU_PLACEMENT=<n>: option "placement" of every context"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hash_join_codes_knl_amd as H

mode = sys.argv[1] if len(sys.argv) > 1 else "serial"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
algo = sys.argv[3] if len(sys.argv) > 3 else "phj"
inner, outer = 64_000_000, 1_000_000_000
base = H.HjGpu(0)
ik, iv, ok, ov = base.column(inner), base.column(inner), base.column(outer), base.column(outer)
base.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
sums = base.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
want = (outer, sums[0], sums[1], sums[2])
keep = []
for i in range(n):
    c = H.HjGpu(0)
    if os.environ.get("HJGPU_PLACEMENT"):
        c.set_option("placement", os.environ["HJGPU_PLACEMENT"])
    best = None
    for _ in range(4):
        assert getattr(c, algo)(ik, iv, inner, ok, ov, outer) == want
        st = c.stats()
        key = "ms_scatter1" if algo != "npj" else "ms_total"
        if best is None or st[key] < best[key]:
            best = st
    print("%s context %d: scatter1 %.3f scatter2 %.3f hist %.3f join %.3f build %.3f total %.3f reserve %.1f"
          % (mode, i, best["ms_scatter1"], best["ms_scatter2"], best["ms_histogram"], best["ms_join"], best["ms_build"], best["ms_total"], best["ms_reserve"]), flush=True)
    if mode == "stacked" and i < 6:
        keep.append(c)
    else:
        c.close()
