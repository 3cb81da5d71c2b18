#!/usr/bin/env python3
"""A/B of join-kernel geometries (option "join_cfg" = "block,log2slots,batch") and of the materialising emit
(option "emit_vec") INSIDE ONE PROCESS, interleaved, on the headline relations: ms of K7+K8 in aggregate mode and
with the three result columns materialised.  Every join is checked against the analytic aggregates.
usage: python tools/ab_join.py [--cfgs 512,13,2 384,13,4 ...] [--rounds R] [--no-rows]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", nargs="+", default=["512,13,2", "512,13,3", "384,13,3", "384,13,4", "256,13,4", "256,13,8"])
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--no-rows", action="store_true")
    a = ap.parse_args()
    import hash_join_codes_knl_amd as H
    hj = H.HjGpu(0)
    inner, outer = a.inner, a.outer
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
    want = (outer, sums[0], sums[1], sums[2])
    hj.reserve(inner, outer)
    block = 16384
    cap = ((outer + block - 1) // block + 4096 + 8) * block
    out = None
    if not a.no_rows:
        jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
        out = (jk, jo, ji, cap, block)
    agg = {c: [] for c in a.cfgs}
    total = {c: [] for c in a.cfgs}
    modes = [(0, 0), (1, 0), (0, 1), (1, 1)]              # (emit_vec, emit_pipe)
    rows = {(c, e): [] for c in a.cfgs for e in modes}
    gaps = {(c, e): [] for c in a.cfgs for e in modes}
    for rnd in range(a.rounds + 1):
        for c in a.cfgs:
            hj.set_option("join_cfg", c)
            for _ in range(2):
                assert hj.phj(ik, iv, inner, ok, ov, outer) == want, c
                st = hj.stats()
                if rnd:
                    agg[c].append(st["ms_join"])
                    total[c].append(st["ms_total"])
            if out:
                for e in modes:
                    hj.set_option("emit_vec", e[0])
                    hj.set_option("emit_pipe", e[1])
                    assert hj.phj(ik, iv, inner, ok, ov, outer, out=out) == want, (c, e)
                    st = hj.stats()
                    assert hj.column_sums(jk, outer, 1, 1)[0] == want[1], (c, e)      # the rows themselves (resets the stats)
                    if rnd:
                        rows[(c, e)].append(st["ms_join"])
                        gaps[(c, e)].append(st["ms_close_gaps"])
                hj.set_option("emit_vec", 1)
                hj.set_option("emit_pipe", 1)
    if out:
        assert hj.column_sums(jk, outer, 1, 1)[0] == want[1]
    rw = 8 * (inner + outer) + 12 * outer
    for c in a.cfgs:
        line = "join_cfg=%-9s aggregate: join med %.3f min %.3f (%.3f of 8 TB/s), step med %.3f" % (
            c, statistics.median(agg[c]), min(agg[c]), 8 * (inner + outer) / (statistics.median(agg[c]) * 1e-3) / 8e12,
            statistics.median(total[c]))
        if out:
            for e in modes:
                m = statistics.median(rows[(c, e)])
                g = statistics.median(gaps[(c, e)])
                line += "\n    rows emit_vec=%d emit_pipe=%d: join med %.3f min %.3f + gaps %.3f (r+w %.3f of 8 TB/s)" % (
                    e[0], e[1], m, min(rows[(c, e)]), g, rw / ((m + g) * 1e-3) / 8e12)
        print(line, flush=True)


if __name__ == "__main__":
    main()
