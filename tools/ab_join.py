#!/usr/bin/env python3
"""A/B of join-kernel geometries (option "join_cfg" = "block,log2slots,batch") INSIDE ONE PROCESS, interleaved, on the
headline relations: ms of K7+K8 in aggregate mode and with the three result columns materialised, plus the
materialising join at several block sizes of the output protocol (npj.cpp:244-246).  Every join is checked against the
analytic aggregates, the materialised keys against their column sum.
usage: python tools/ab_join.py [--cfgs 512,13,2 512,13,4 ...] [--rounds R] [--no-rows] [--blocks 1024 4096 ...]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", nargs="+", default=["512,13,2", "256,12,2"])
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--no-rows", action="store_true")
    ap.add_argument("--blocks", type=int, nargs="*", default=[], help="also time the materialising join with these block sizes")
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
    cap = ((outer + block - 1) // block + 4096 + 8) * block + (8192 + 8) * 65536
    out = None
    if not a.no_rows:
        jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
        out = (jk, jo, ji, cap, block)
    agg = {c: [] for c in a.cfgs}
    total = {c: [] for c in a.cfgs}
    rows = {c: [] for c in a.cfgs}
    gaps = {c: [] for c in a.cfgs}
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
                assert hj.phj(ik, iv, inner, ok, ov, outer, out=out) == want, c
                st = hj.stats()
                assert hj.column_sums(jk, outer, 1, 1)[0] == want[1], c          # the rows themselves (resets the stats)
                if rnd:
                    rows[c].append(st["ms_join"])
                    gaps[c].append(st["ms_close_gaps"])
    rw = 8 * (inner + outer) + 12 * outer
    for bs in a.blocks if out else []:
        capb = ((outer + bs - 1) // bs + 8192 + 8) * bs
        if capb > cap:
            print("block %d: needs %d rows of capacity, have %d" % (bs, capb, cap))
            continue
        hj.set_option("join_cfg", a.cfgs[0])
        tj, tg = [], []
        for _ in range(4):
            assert hj.phj(ik, iv, inner, ok, ov, outer, out=(jk, jo, ji, capb, bs)) == want
            st = hj.stats()
            tj.append(st["ms_join"])
            tg.append(st["ms_close_gaps"])
        print("block_size %6d: join med %.3f + gaps %.3f (r+w %.3f of 8 TB/s)" % (
            bs, statistics.median(tj[1:]), statistics.median(tg[1:]),
            rw / ((statistics.median(tj[1:]) + statistics.median(tg[1:])) * 1e-3) / 8e12), flush=True)
    for c in a.cfgs:
        line = "join_cfg=%-9s aggregate: join med %.3f min %.3f (%.3f of 8 TB/s), step med %.3f" % (
            c, statistics.median(agg[c]), min(agg[c]), 8 * (inner + outer) / (statistics.median(agg[c]) * 1e-3) / 8e12,
            statistics.median(total[c]))
        if out:
            m, g = statistics.median(rows[c]), statistics.median(gaps[c])
            line += " | rows: join med %.3f min %.3f + gaps %.3f (r+w %.3f of 8 TB/s)" % (m, min(rows[c]), g, rw / ((m + g) * 1e-3) / 8e12)
        print(line, flush=True)


if __name__ == "__main__":
    main()
