#!/usr/bin/env python3
"""What bounds the NPJ build (K2, npj.cpp:190-212)?  hjgpu_random_cas_ms = 64 M independent pseudo-random 8-byte CAS into a
zeroed 2 GiB buffer (the build's shape: table of 64 M / 0.25 buckets), 1 / 2 / 4 / 8 in flight per lane, with and without a
look at the bucket first; next to it the build itself.  (Round 4 also built a K2 with 1 / 2 / 4 / 8 claims in flight per lane
that CASes without looking first: 4.98 / 14.2 / 37.5 / 37.4 ms - with the line-hashed table every key of a line starts at the
line's first bucket, so a blind CAS fails for every second key; measured and removed, profiles/r04_npj_build_ceiling.txt.)
usage: python tools/npj_build_ceiling.py [--inner 64000000 --outer 1000000000]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    a = ap.parse_args()
    import numpy as np
    import hash_join_codes_knl_amd as H
    hj = H.HjGpu(0)
    table_bytes = 2 << 30
    buf = hj.column(table_bytes // 8, np.uint64)
    print("library %s; random CAS ceiling: %d ops into %d MiB" % (H.kernel_hash(), a.inner, table_bytes >> 20))
    for load in (0, 1):
        for u in (1, 2, 4, 8):
            ms = min(hj.random_cas_ms(buf, table_bytes, a.inner, u, load) for _ in range(3))
            print("  %d in flight per lane, %s: %.3f ms = %.1f G CAS/s" % (u, "look first" if load else "CAS only  ", ms, a.inner / ms / 1e6), flush=True)
    buf.free()
    ik, iv, ok, ov = hj.column(a.inner), hj.column(a.inner), hj.column(a.outer), hj.column(a.outer)
    hj.generate(1, a.inner, a.outer, 0, a.outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, a.outer, 0x9E3779B1, 0x2545F491)
    want = (a.outer, sums[0], sums[1], sums[2])
    best = None
    for _ in range(5):
        got = hj.npj(ik, iv, a.inner, ok, ov, a.outer)
        assert tuple(got) == want, (got, want)
        st = hj.stats()
        best = st if best is None or st["ms_build"] < best["ms_build"] else best
    print("NPJ %d x %d: build (table clear + claims) %.3f ms, probe %.3f ms, total %.3f ms"
          % (a.inner, a.outer, best["ms_build"], best["ms_join"], best["ms_total"]), flush=True)
    hj.close()


if __name__ == "__main__":
    main()
