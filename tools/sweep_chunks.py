import sys, os, statistics
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import hash_join_codes_knl_amd as H
hj = H.HjGpu(0)
inner, outer = 64_000_000, 1_000_000_000
ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
sums = hj.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
want = (outer, sums[0], sums[1], sums[2])
for rnd in range(2):
    for c in (1, 2, 4, 8):
        ts = []
        for _ in range(4):
            assert hj.cpra(ik, iv, inner, ok, ov, outer, H.PhjParams(chunks=c)) == want
            ts.append(hj.stats())
        b = min(ts, key=lambda s: s["ms_total"])
        print("chunks %d: total %.3f hist %.3f plan %.3f scatter1 %.3f scatter2 %.3f join %.3f" % (c, b["ms_total"], b["ms_histogram"], b["ms_plan"], b["ms_scatter1"], b["ms_scatter2"], b["ms_join"]), flush=True)
