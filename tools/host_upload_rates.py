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

# the materialising host call (three result columns of |S| rows into page-locked host columns): rows go home per batch
# behind the upload (option host_batch), or after the join (host_batch = 0); the C entry point directly, columns made once
if len(sys.argv) > 2 and sys.argv[2] == "rows":
    import ctypes as C
    from hash_join_codes_knl_amd import api
    cap = outer + 4096
    out = [hj.host_column(cap) for _ in range(3)]
    rows = api.HostRows(out[0].array.ctypes.data, out[1].array.ctypes.data, out[2].array.ctypes.data, cap)
    a = [p.array for p in pinned]
    for hb in (64 << 20, 64 << 20, 64 << 20, 0, 64 << 20):
        hj.set_option("host_batch", hb)
        r, st = api.Result(), api.Stats()
        t0 = time.perf_counter()
        rc = hj.lib.hjgpu_join_host_rows(hj.handle, 1, a[0].ctypes.data, a[1].ctypes.data, a[0].size, a[2].ctypes.data, a[3].ctypes.data,
                                         a[2].size, None, None, C.byref(rows), C.byref(r), C.byref(st))
        wall = time.perf_counter() - t0
        assert rc == 0 and r.as_tuple() == want, (rc, r.as_tuple())
        print("rows, host_batch %9d: upload %.1f ms, download %.1f ms, batches %d, call %.1f ms -> %.2f Gtuples/s PCIe-inclusive"
              % (hb, st.ms_upload, st.ms_download, st.batches, wall * 1e3, outer / wall / 1e9), flush=True)
    assert int(out[0].array[:outer].astype(np.uint64).sum()) == want[1]
