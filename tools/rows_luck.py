#!/usr/bin/env python3
"""Does the materialising join's time depend on WHICH allocation its three result columns live in (as K6 pass 1's does on
its twin)?  Fresh result columns per round (the old ones kept alive, so every round gets other memory), one context."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hash_join_codes_knl_amd as H

hj = H.HjGpu(0)
inner, outer = 64_000_000, 1_000_000_000
ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
sums = hj.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
want = (outer, sums[0], sums[1], sums[2])
block = 4096
cap = ((outer + block - 1) // block + 4096 + 8) * block
keep = []
placed = len(sys.argv) > 1 and sys.argv[1] == "placed"
for rnd in range(8):
    if len(sys.argv) > 1 and sys.argv[1] in ("oneblock", "oneblock_placed"):
        # ONE allocation for the three columns (the two kinds of allocations are sharp for blocks of 8 GB and more)
        blk = hj.column(3 * cap, placed=sys.argv[1] == "oneblock_placed")
        class View:
            def __init__(self, ptr): self.ptr = ptr
            def free(self): pass
        cols = [View(blk.ptr + 4 * cap * i) for i in range(3)]
        cols[0].free = blk.free
    else:
        cols = [hj.column(cap, placed=placed) for _ in range(3)]
    best = None
    for _ in range(3):
        assert hj.phj(ik, iv, inner, ok, ov, outer, out=(*(c.ptr for c in cols), cap, block)) == want
        st = hj.stats()
        if best is None or st["ms_join"] < best["ms_join"]:
            best = st
    print((sys.argv[1] + " " if len(sys.argv) > 1 else "") + "result columns %d: join %.3f + gaps %.3f ms, scatter1 %.3f" % (rnd, best["ms_join"], best["ms_close_gaps"], best["ms_scatter1"]), flush=True)
    keep.append(cols)
    if len(keep) > 4:
        for c in keep.pop(0):
            c.free()
