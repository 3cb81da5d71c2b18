"""How fast are random table lookups when the table fits the caches?  NPJ probe of 1 G tuples against
build sides from 50 K to 16 M tuples (tables of 1.6 MB ... 512 MB at load 0.25): the L2-resident rate is what a
'one partitioning pass + L2-resident tables' join design would get for its probe phase."""
import os, sys
sys.path.insert(0, os.getcwd())
import hash_join_codes_knl_amd as H
hj = H.HjGpu(0)
outer = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
for inner in (50_000, 100_000, 200_000, 400_000, 1_000_000, 4_000_000, 16_000_000):
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    for load in (0.25, 0.5):
        best = 1e9
        for rep in range(3):
            got = hj.npj(ik, iv, inner, ok, ov, outer, H.NpjParams(load=load))
            st = hj.stats()
            best = min(best, st["ms_join"])
        print("inner %9d load %.2f table %7.1f MB  probe %.3f ms  (%.0f G lookups/s) count ok %s" %
              (inner, load, st["buckets"] * 8 / 1e6, best, outer / best / 1e6, got[0] == outer), flush=True)
    for c in (ik, iv, ok, ov):
        c.free()
