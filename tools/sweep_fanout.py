#!/usr/bin/env python3
"""A/B timing of (fanout1, fanout2[, env]) choices INSIDE ONE PROCESS (same device, same
buffers; boxes and even processes differ by +-10 %).
usage: python tools/sweep_fanout.py 136x136 512x37 "512x37:scatter_cfg=1024,2,0" ...   (options: hjgpu_set_option)
       [--inner N --outer N --rounds R --reps K]
Every run is checked against the analytic aggregates."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="+")
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import hash_join_codes_knl_amd as H
    hj = H.HjGpu(0)
    hj.set_option("solo", "1")               # blocking joins of a process that runs nothing else: partial-line stores plain (DESIGN section 3 "Round 5")
    ik, iv, ok, ov = hj.column(a.inner), hj.column(a.inner), hj.column(a.outer), hj.column(a.outer)
    hj.generate(1, a.inner, a.outer, 0, a.outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, a.outer, 0x9E3779B1, 0x2545F491)
    want = (a.outer, sums[0], sums[1], sums[2])
    phases = ["ms_total", "ms_histogram", "ms_plan", "ms_scatter1", "ms_scatter2", "ms_join"]
    data = {c: {p: [] for p in phases} for c in a.cases}
    touched = set()
    for rnd in range(a.rounds):
        for case in a.cases:
            spec, _, envs = case.partition(":")
            f1, f2 = (int(x) for x in spec.split("x"))
            for k in touched:
                hj.set_option(k, "" if k.endswith("_cfg") else 0)
            for kv in filter(None, envs.split(";")):
                k, v = kv.split("=", 1)
                k = k.lower().replace("hjgpu_", "")
                hj.set_option(k, v)
                touched.add(k)
            prm = H.PhjParams(fanout1=f1, fanout2=f2)
            for _ in range(a.reps):
                got = hj.phj(ik, iv, a.inner, ok, ov, a.outer, prm)
                assert got == want, (case, got, want)
                st = hj.stats()
                if rnd > 0 or a.rounds == 1:
                    for p in phases:
                        data[case][p].append(st[p])
    for case in a.cases:
        print("%-44s" % case, " ".join("%s %.3f/%.3f |" % (p[3:], statistics.median(x), min(x))
                                         for p, x in data[case].items()), flush=True)


if __name__ == "__main__":
    main()
