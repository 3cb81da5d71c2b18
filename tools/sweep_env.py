#!/usr/bin/env python3
"""A/B timing of library options (hjgpu_set_option) INSIDE ONE PROCESS (same device, same
buffers): boxes differ by +-10 % and even processes on one box differ, so only
interleaved rounds in one process are comparable (cdna_hip_programming.md rule 24).

usage: python tools/sweep_env.py OPTION v1 v2 ...   (e.g. dense2 0 1; scatter_cfg 1024,4 1024,3; "" resets a cfg)
       python tools/sweep_env.py OPTION v1 v2 ... [--inner N --outer N --rounds R --algo phj]
Prints the median / min of every phase per value."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("var")
    ap.add_argument("values", nargs="+")
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--algo", default="phj")
    a = ap.parse_args()
    import hash_join_codes_knl_amd as H
    hj = H.HjGpu(0)
    ik, iv, ok, ov = hj.column(a.inner), hj.column(a.inner), hj.column(a.outer), hj.column(a.outer)
    hj.generate(1, a.inner, a.outer, 0, a.outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, a.outer, 0x9E3779B1, 0x2545F491)
    want = (a.outer, sums[0], sums[1], sums[2])
    fn = getattr(hj, a.algo)
    phases = ["ms_total", "ms_histogram", "ms_plan", "ms_scatter1", "ms_scatter2", "ms_join", "ms_build"]
    data = {v: {p: [] for p in phases} for v in a.values}
    for rnd in range(a.rounds):
        for v in a.values:
            hj.set_option(a.var.lower().replace("hjgpu_", ""), v)
            for _ in range(a.reps):
                got = fn(ik, iv, a.inner, ok, ov, a.outer)
                assert got == want, (v, got, want)
                st = hj.stats()
                if rnd > 0 or a.rounds == 1:          # round 0 warms caches / allocations
                    for p in phases:
                        data[v][p].append(st[p])
    for v in a.values:
        print("%s=%-6s" % (a.var, v), " ".join("%s med %.3f min %.3f |" % (p[3:], statistics.median(x), min(x))
                                               for p, x in data[v].items() if max(x) > 0))


if __name__ == "__main__":
    main()
