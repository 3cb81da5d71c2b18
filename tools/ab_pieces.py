#!/usr/bin/env python3
"""The receiving side of the multi-GPU CPRA at full size on one GPU: R and S arrive as 8 pass-1-partitioned pieces (every
"sender" partitions 1/8 of the relation with hjgpu_partition_packed_async, the pieces lie one after the other as in a
receive buffer), the build side is prepared once (hjgpu_phj_build_prepartitioned) and the probe side joined in one batch.
A/B of option "piece_interleave" (pass 2 takes its tiles partition by partition across the pieces, or piece by piece),
interleaved in one process.  usage: python tools/ab_pieces.py [--pieces 8 --rounds 4]"""
import argparse
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pieces", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    a = ap.parse_args()
    import hash_join_codes_knl_amd as H
    hj, part = H.HjGpu(0), H.HjGpu(0)
    F1, f1 = 192, 0x2C1B3C6D
    inner, outer = a.inner, a.outer
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
    want = [outer, sums[0], sums[1], sums[2]]

    def arrive(keys, vals, n):
        """n rows as `pieces` pass-1-partitioned pieces, back to back; returns (tuples, piece offsets)"""
        buf, off = hj.column(n + 64, np.uint64), hj.column(F1 + 1, np.uint64)
        per = (n // a.pieces) & ~15
        offs = [0]
        for c in range(a.pieces):
            b, e = c * per, n if c + 1 == a.pieces else (c + 1) * per
            assert b % 16 == 0
            part.partition_packed_async(keys.ptr + 4 * b, vals.ptr + 4 * b, e - b, f1, F1, buf.ptr + 8 * b, off)
            part.synchronize()
            offs.append(e)
        return buf, offs

    rt, roffs = arrive(ik, iv, inner)
    st_, soffs = arrive(ok, ov, outer)
    d_res = hj.column(4, np.uint64)
    t = {0: [], 1: []}
    for rnd in range(a.rounds + 1):
        for opt in (0, 1):
            hj.set_option("piece_interleave", opt)
            hj.phj_build_prepartitioned(rt, hj.prepartitioned(f1, F1, 0, F1, roffs), outer)
            hj.phj_probe_prepartitioned_async(st_, hj.prepartitioned(f1, F1, 0, F1, soffs), d_res)
            hj.get_async_status()
            assert [int(x) for x in d_res.download()] == want, opt
            s = hj.stats()
            if rnd:
                t[opt].append((s["ms_total"], s["ms_histogram"], s["ms_scatter2"], s["ms_join"]))
    for opt in (0, 1):
        m = [statistics.median(x[i] for x in t[opt]) for i in range(4)]
        print("piece_interleave=%d (%d pieces): probe batch total %.3f ms, K4p %.3f, pass 2 %.3f, join %.3f" % (opt, a.pieces, *m), flush=True)


if __name__ == "__main__":
    main()
