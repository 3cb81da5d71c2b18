import os, sys, statistics
sys.path.insert(0, os.getcwd())
import hash_join_codes_knl_amd as H
hj = H.HjGpu(0)
inner, outer = 64_000_000, 1_000_000_000
ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
ph = ["ms_histogram","ms_scatter1","ms_scatter2","ms_join"]
d = {p: [] for p in ph}
for i in range(8):
    hj.phj(ik, iv, inner, ok, ov, outer)
    st = hj.stats()
    if i >= 2:
        for p in ph: d[p].append(st[p])
print(" ".join("%s %.3f" % (p[3:], statistics.median(v)) for p, v in d.items()))
