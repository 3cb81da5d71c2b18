#!/usr/bin/env python3
"""PCIe-inclusive rate of hjgpu_join_host at |R| = 64 M, |S| = 1 G: page-locked columns
(hjgpu_host_alloc, what the host programs use) against pageable numpy columns."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hash_join_codes_knl_amd as H

inner, outer = 64_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
hj = H.HjGpu(0)
d = [hj.column(n) for n in (inner, inner, outer, outer)]
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, *d)
pinned = [hj.host_column(n) for n in (inner, inner, outer, outer)]
for dst, src in zip(pinned, d):
    hj.lib.hjgpu_memcpy_d2h(hj.handle, dst.ptr, src.ptr, 4 * len(dst.array))
sums = hj.column_sums(d[2], outer, 0x9E3779B1, 0x2545F491)
want = (outer, sums[0], sums[1], sums[2])
for c in d:
    c.free()
for label, cols in (("pinned", pinned), ("pageable", [np.array(p.array) for p in pinned])):
    for algo, name in ((1, "phj"), (0, "npj")):
        t0 = time.perf_counter()
        got, st = hj.join_host(algo, *cols)
        wall = time.perf_counter() - t0
        assert got == want, (label, name)
        gb = 8.0 * (inner + outer) / 1e9
        print("%-8s %s: upload %.1f ms (%.1f GB/s), device join %.1f ms, call %.1f ms -> %.2f Gtuples/s PCIe-inclusive"
              % (label, name, st["ms_upload"], gb / st["ms_upload"] * 1e3, st["ms_total"], wall * 1e3, outer / wall / 1e9), flush=True)
