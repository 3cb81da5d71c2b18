import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import hash_join_codes_knl_amd as H
inner, outer = 64_000_000, 1_000_000_000
hj = H.HjGpu(0)
d = [hj.column(n) for n in (inner, inner, outer, outer)]
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, *d)
pinned = [hj.host_column(n) for n in (inner, inner, outer, outer)]
for dst, src in zip(pinned, d):
    hj.lib.hjgpu_memcpy_d2h(hj.handle, dst.ptr, src.ptr, 4 * len(dst.array))
for c in d:
    c.free()
for i in range(4):
    t0 = time.perf_counter()
    got, st = hj.join_host(1, *pinned)
    wall = time.perf_counter() - t0
    print("call %d: upload %.1f ms, device join %.2f ms in %d batches, reserve so far %.1f ms, wall %.1f ms" % (i, st["ms_upload"], st["ms_total"], st["batches"], st["ms_reserve"], wall * 1e3), flush=True)
