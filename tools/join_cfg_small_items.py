import os, sys
sys.path.insert(0, os.getcwd())
import hash_join_codes_knl_amd as H
fi, fo = 0x2545F491, 0x9E3779B1
for inner, outer in ((1_000_000_000, 1_000_000_000), (1_000_000_000, 4_000_000_000), (48_000_000, 48_000_000), (48_000_000, 768_000_000)):
    with H.HjGpu(0) as hj:
        ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
        hj.generate(3, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
        sums = hj.column_sums(ok, outer, fo, fi)
        want = (outer, sums[0], sums[1], sums[2])
        hj.set_option("group_from", "300000000"); hj.set_option("group_always", "1")
        for cfg, per in (("512,13,2", 64_000_000), ("512,13,2", 48_000_000), ("256,12,2", 48_000_000), ("256,12,2", 32_000_000)):
            hj.set_option("join_cfg", cfg); hj.set_option("group_inner", str(per))
            best = None
            for rep in range(3):
                got = hj.phj(ik, iv, inner, ok, ov, outer)
                st = hj.stats()
                if best is None or st["ms_total"] < best["ms_total"]:
                    best = st
            print("PHJ %d x %d join_cfg %s group_inner %d: %s total %.3f scatter0 %.3f hist %.3f plan %.3f sc1 %.3f sc2 %.3f join %.3f groups %d fan %dx%d"
                  % (inner, outer, cfg, per, "ok" if got == want else "MISMATCH", best["ms_total"], best["ms_scatter0"], best["ms_histogram"], best["ms_plan"],
                     best["ms_scatter1"], best["ms_scatter2"], best["ms_join"], best["groups"], best["fanout1"], best["fanout2"]), flush=True)
