"""hjgpu_partition (one pass, separate key / payload columns in and out) over 1 G tuples at small fan-outs:
what the exchange-level partitioning of the multi-GPU CPRA path costs per GPU."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import hash_join_codes_knl_amd as H
hj = H.HjGpu(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ik, iv, ok, ov = hj.column(16), hj.column(16), hj.column(n), hj.column(n)
hj.generate(1, 16, n, 0, n, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
pk, pv = hj.column(n), hj.column(n)
for F in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
    off = hj.column(F + 1, np.uint64)
    best = 1e9
    for rep in range(3):
        hj.synchronize()
        t0 = time.perf_counter()
        hj.partition(ok, ov, n, 0x2C1B3C6D | 1, F, pk, pv, off)
        hj.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print("fan-out %5d: %.3f ms wall (%.1f GB/s read + written)" % (F, best, 16.0 * n / best / 1e6), flush=True)
    off.free()
