#!/usr/bin/env python3
"""NPJ probe time vs hash-table load factor (results do not depend on it; the
reference uses 0.90, npj.cpp:944).  One process, same buffers."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hash_join_codes_knl_amd as H

hj = H.HjGpu(0)
inner, outer = 64_000_000, 1_000_000_000
ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
sums = hj.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
want = (outer, sums[0], sums[1], sums[2])
for load in [float(x) for x in (sys.argv[1:] or ["0.25", "0.5", "0.6", "0.7", "0.8", "0.9"])]:
    b, j = [], []
    for i in range(4):
        assert hj.npj(ik, iv, inner, ok, ov, outer, H.NpjParams(load=load)) == want
        st = hj.stats()
        if i:
            b.append(st["ms_build"]); j.append(st["ms_join"])
    print("load %.2f  buckets %d  build %.2f ms  probe %.2f ms" % (load, st["buckets"], statistics.median(b), statistics.median(j)))
