#!/usr/bin/env python3
"""Does the allocation that holds the caller's INPUT columns change the time of K6 pass 1 (as the allocation that holds its
output twin does)?  One context (its workspace is made once and stays), N sets of probe-side columns held side
by side (each a fresh hipMalloc), the same data copied into every set, the same join run on every set.
usage: python tools/input_luck.py [sets 8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hash_join_codes_knl_amd as H

sets = int(sys.argv[1]) if len(sys.argv) > 1 else 8
inner, outer = 64_000_000, 1_000_000_000
hj = H.HjGpu(0)
print("# kernel hash", H.kernel_hash(), flush=True)
ik, iv = hj.column(inner), hj.column(inner)
cols = [(hj.column(outer), hj.column(outer)) for _ in range(sets)]
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, cols[0][0], cols[0][1])
sums = hj.column_sums(cols[0][0], outer, 0x9E3779B1, 0x2545F491)
want = (outer, sums[0], sums[1], sums[2])
for k, v in cols[1:]:
    hj.memcpy_d2d(k, cols[0][0], 4 * outer) if hasattr(hj, "memcpy_d2d") else None
for rnd in range(2):
    for i, (k, v) in enumerate(cols):
        if i and rnd == 0:
            # no device-to-device copy in the C-ABI: through the generator again (same seed, same data)
            hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, k, v)
        best = None
        for _ in range(4):
            assert hj.phj(ik, iv, inner, k, v, outer) == want
            st = hj.stats()
            if best is None or st["ms_scatter1"] < best["ms_scatter1"]:
                best = st
        print("round %d input set %d (keys at %#x): hist %.3f scatter1 %.3f scatter2 %.3f join %.3f total %.3f"
              % (rnd, i, k.ptr, best["ms_histogram"], best["ms_scatter1"], best["ms_scatter2"], best["ms_join"], best["ms_total"]), flush=True)
