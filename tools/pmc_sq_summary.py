#!/usr/bin/env python3
"""profiles/<tag>_pmc_sq.csv (tools/pmc_sq.sh: kernel, dispatch, counter, value) -> the per-kernel summary DESIGN section 5
argues from: for every shipped kernel of the headline PHJ its PROBE-SIDE launch (the dispatch with the most vector
instructions), wave-instructions per 64 tuples, LDS bank-conflict share, and for K6 the cycles per 16 384-tuple tile.
usage: python tools/pmc_sq_summary.py profiles/r03_pmc_sq.csv [--outer 1000000000 --inner 64000000] > profiles/r03_pmc_sq_summary.txt"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--ghz", type=float, default=2.4)
    ap.add_argument("--cus", type=int, default=256)
    a = ap.parse_args()
    raw = collections.defaultdict(dict)
    for r in csv.DictReader(open(a.csv)):
        raw[(r["kernel"], int(r["dispatch"]))][r["counter"]] = float(r["value"])
    # the two counter sets come from two runs of the same program: the n-th launch of a kernel in one is the n-th in the
    # other (dispatch ids may differ); launches are merged by their order
    d = {}
    kernels = sorted({k for k, _ in raw})
    for k in kernels:
        first = sorted(disp for (kk, disp), c in raw.items() if kk == k and "SQ_INSTS_VALU" in c)
        second = sorted(disp for (kk, disp), c in raw.items() if kk == k and "SQ_ACTIVE_INST_VALU" in c)
        for n, disp in enumerate(first):
            c = dict(raw[(k, disp)])
            if n < len(second):
                c.update(raw[(k, second[n])])
            d[(k, str(disp))] = c
    best = {}
    for (k, disp), c in d.items():
        # the aggregate-mode launch: the materialising join of bench.py's `materialized` leg writes three result columns
        if "join" in k and c.get("SQ_INSTS_VMEM_WR", 0) > 0.01 * c.get("SQ_INSTS_VALU", 1):
            continue
        if k not in best or c.get("SQ_INSTS_VALU", 0) > d[(k, best[k])].get("SQ_INSTS_VALU", 0):
            best[k] = disp
    print("# SQ counters of the shipped kernels (tools/pmc_sq.sh: rocprofv3 --pmc in two passes over `bench.py --steps 1 --warmup 1`),")
    print("# the PROBE-SIDE launch of every kernel of the headline PHJ (|R| = %d x |S| = %d; %.3f M wave-rows of 64 tuples;" % (
        a.inner, a.outer, a.outer / 64e6))
    print("# join: %.3f M incl. the build side).  wave-instructions per 64 tuples, LDS bank-conflict share, cycles per 16 384-tuple tile." % (
        (a.outer + a.inner) / 64e6))
    print("# made by tools/pmc_sq_summary.py from %s" % a.csv)
    for k in sorted(best, key=lambda k: ("hist" not in k, "scatter" not in k, k)):
        c = d[(k, best[k])]
        rows = (a.outer + (a.inner if "join" in k else 0)) / 64.0
        per = lambda name: c.get(name, 0.0) / rows
        print()
        print("%s   (dispatch %s, SQ_WAVES %d)" % (k, best[k], c.get("SQ_WAVES", 0)))
        print("  per 64 tuples: VALU %.1f  SALU %.1f  LDS %.2f  VMEM read %.2f  VMEM write %.2f wave-instructions" % (
            per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), per("SQ_INSTS_VMEM_RD"), per("SQ_INSTS_VMEM_WR")))
        idx = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        print("  LDS: bank-conflict cycles / index-active cycles = %.2f; SQ_WAIT_INST_LDS %.0f M wave-cycles; SQ_BUSY_CYCLES %.1f M" % (
            c.get("SQ_LDS_BANK_CONFLICT", 0.0) / idx if idx else 0.0, c.get("SQ_WAIT_INST_LDS", 0.0) / 1e6, c.get("SQ_BUSY_CYCLES", 0.0) / 1e6))
        if "scatter" in k:
            tiles_per_cu = a.outer / 16384.0 / a.cus
            # a wave-instruction occupies its SIMD for 4 cycles; 4 SIMDs and one scalar unit per CU; the LDS counters are per CU already
            valu = c.get("SQ_INSTS_VALU", 0.0) * 4 / (a.cus * 4) / tiles_per_cu
            salu = c.get("SQ_INSTS_SALU", 0.0) / a.cus / tiles_per_cu
            ldsi = idx / a.cus / tiles_per_cu
            conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / a.cus / tiles_per_cu
            print("  per tile and CU (16 waves, 256 wave-rows): VALU %.0f cycles per SIMD, SALU %.0f issue cycles, LDS index-active %.0f cycles "
                  "(of them conflicts %.0f)" % (valu, salu, ldsi, conf))


if __name__ == "__main__":
    main()
