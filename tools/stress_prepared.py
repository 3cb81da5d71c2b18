#!/usr/bin/env python3
"""Single-GPU stress of a prepared build side probed by back-to-back enqueue-only batches (what the multi-GPU CPRA does
per rank), optionally with another context partitioning on a second stream at the same time.
usage: python tools/stress_prepared.py [--steps 30 --slices 8 --concurrent 0|1 --rebuild 0|1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--slices", type=int, default=8)
    ap.add_argument("--concurrent", type=int, default=0)
    ap.add_argument("--rebuild", type=int, default=1)
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    a = ap.parse_args()
    import torch
    torch.cuda.init()
    import hash_join_codes_knl_amd as H
    hj, other = H.HjGpu(0), H.HjGpu(0)
    inner, outer = a.inner, a.outer
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
    want = [outer, sums[0], sums[1], sums[2]]
    per = (outer // a.slices) & ~15
    cuts = [(i * per, outer if i + 1 == a.slices else (i + 1) * per) for i in range(a.slices)]
    d_res = [hj.column(4, np.uint64) for _ in range(a.slices)]
    s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
    tk, tv, to = other.column(per + 64), other.column(per + 64), other.column(9, np.uint64)
    max_outer = max(e - b for b, e in cuts)
    hj.phj_build(ik, iv, inner, max_outer, None, s_main.cuda_stream)
    bad = 0
    for step in range(a.steps):
        if a.rebuild:
            hj.phj_build(ik, iv, inner, max_outer, None, s_main.cuda_stream)
        for i, (b, e) in enumerate(cuts):
            if a.concurrent:      # the exchange-level partitioning of the next slice, another context, another stream
                other.partition_async(ok.ptr + 4 * b, ov.ptr + 4 * b, e - b, 0x2C1B3C6D, 8, tk, tv, to, s_side.cuda_stream)
            hj.phj_probe_async(ok.ptr + 4 * b, ov.ptr + 4 * b, e - b, d_res[i], s_main.cuda_stream)
        torch.cuda.synchronize()
        tot = [0, 0, 0, 0]
        for r in d_res:
            tot = [(x + int(y)) & ((1 << 64) - 1) for x, y in zip(tot, r.download())]
        if tot != want:
            bad += 1
            print("step %d WRONG: count %+d" % (step, tot[0] - want[0]), flush=True)
    print("slices %d concurrent %d rebuild %d: %d of %d steps wrong" % (a.slices, a.concurrent, a.rebuild, bad, a.steps), flush=True)


if __name__ == "__main__":
    main()
