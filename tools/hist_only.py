"""hjgpu_histogram alone (K4 with nothing downstream): wall time per call at several fan-outs over 1 G keys."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import hash_join_codes_knl_amd as H
hj = H.HjGpu(0)
n = 1_000_000_000
ik, iv, ok, ov = hj.column(16), hj.column(16), hj.column(n), hj.column(n)
hj.generate(1, 16, n, 0, n, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
for F in (128, 1024, 4096, 18432, 32768):
    cnt = hj.column(F, np.uint64)
    best = 1e9
    for rep in range(5):
        hj.synchronize()
        t0 = time.perf_counter()
        hj.histogram(ok, n, 0x9E3779B1, F, cnt)
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print("fan-out %6d: %.3f ms, sum of counts %d" % (F, best, int(cnt.download().sum())), flush=True)
    cnt.free()
