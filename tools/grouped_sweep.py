#!/usr/bin/env python3
"""Grouped plans (a third partitioning pass, option group_from / group_inner) against the two-pass plan with multi-fill tables:
PHJ device time for build sides beyond what two passes hold.  usage: python tools/grouped_sweep.py [outer] [inner ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hash_join_codes_knl_amd as H

outer = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
inners = [int(x) for x in sys.argv[2:]] or [256_000_000, 384_000_000, 512_000_000, 1_000_000_000]
fi, fo = 0x2545F491, 0x9E3779B1
print("# kernel hash", H.kernel_hash() if hasattr(H, "kernel_hash") else "?", flush=True)
for inner in inners:
    with H.HjGpu(0) as hj:
        ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
        hj.generate(3, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
        sums = hj.column_sums(ok, outer, fo, fi)
        want = (outer, sums[0], sums[1], sums[2])
        for per in (0, 64_000_000, 100_000_000, 200_000_000):
            hj.set_option("group_from", "0" if per == 0 else "1")
            hj.set_option("group_always", "1")
            if per:
                hj.set_option("group_inner", str(per))
            best = None
            for rep in range(3):
                got = hj.phj(ik, iv, inner, ok, ov, outer)
                st = hj.stats()
                if best is None or st["ms_total"] < best["ms_total"]:
                    best = st
            keys = ("ms_total", "ms_scatter0", "ms_histogram", "ms_plan", "ms_scatter1", "ms_scatter2", "ms_join", "groups", "fanout1", "fanout2")
            print("PHJ %d x %d group_inner %d: %s %s" % (inner, outer, per, "ok" if got == want else "MISMATCH",
                  " ".join("%s %s" % (k[3:] if k.startswith("ms_") else k, round(best[k], 3) if isinstance(best[k], float) else best[k]) for k in keys)), flush=True)
