#!/usr/bin/env python3
"""Rate of independent pseudo-random 64-byte line reads (hjgpu_random_line_read_ms: the NPJ probe's access shape) as a
function of the table size: L2-resident (<= 4 MiB per XCD), Infinity-Cache-resident (<= 256 MiB), HBM.
usage: python tools/ubench_line_sweep.py [--reads N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000_000)
    a = ap.parse_args()
    import hash_join_codes_knl_amd as H
    hj = H.HjGpu(0)
    big = hj.column(2 << 28)                      # 2 GiB
    for kib in (64, 256, 1024, 2048, 4096, 8192, 32768, 131072, 524288, 2097152):
        best = min(hj.random_line_read_ms(big, kib * 1024, a.reads) for _ in range(3))
        print("table %8d KiB: %7.3f ms for %d reads = %6.1f G lines/s = %6.2f TB/s of lines" % (
            kib, best, a.reads, a.reads / best / 1e6, a.reads * 64 / best / 1e9), flush=True)


if __name__ == "__main__":
    main()
