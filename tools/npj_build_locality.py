"""NPJ build (64 M inserts into a 2 GB table) from the build side as given and from the same rows reordered by
table region (one partitioning pass with the table's own hash: mulhi is monotone, so partition p of F covers the
p-th F-th of the table): does locality of the random CAS traffic pay for the extra pass?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import hash_join_codes_knl_amd as H
hj = H.HjGpu(0)
inner, outer = 64_000_000, 64_000_000
f = 0x9E3779B1
ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
buckets = 256_000_000
table = hj.column(buckets, np.uint64)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        hj.synchronize(); t0 = time.perf_counter(); fn(); hj.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
print("build from the given order: %.3f ms" % timed(lambda: hj.npj_build(ik, iv, inner, table, buckets, f)), flush=True)
for F in (64, 256, 1024):
    pk, pv, off = hj.column(inner), hj.column(inner), hj.column(F + 1, np.uint64)
    tp = timed(lambda: hj.partition(ik, iv, inner, f, F, pk, pv, off))
    tb = timed(lambda: hj.npj_build(pk, pv, inner, table, buckets, f))
    print("fan-out %5d: partition %.3f ms + build %.3f ms = %.3f ms" % (F, tp, tb, tp + tb), flush=True)
    for c in (pk, pv, off):
        c.free()
