#!/usr/bin/env python3
"""Repeats hjgpu_join_host_rows - host columns in, the materialised join back in host columns, the probe side in batches behind the
upload (join + row-copy kernel + next upload side by side on three streams) - and checks EVERY step's ROWS on the host: the number
of rows and the sums of the three returned columns against the analytic aggregates (the device's own aggregates are taken before the
rows are stored and compacted: a lost row store or a lost close_gaps move would not change them).  The sums of step s are taken by
worker threads while step s + 1 runs.
usage: python tools/stress_host_rows.py [--algo phj|npj|cpra --steps 200 --inner N --outer N --batch ROWS --pageable]"""
import argparse
import ctypes as C
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="phj", choices=["npj", "phj", "cpra"])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--inner", type=int, default=1_000_000)
    ap.add_argument("--outer", type=int, default=8_000_000)
    ap.add_argument("--batch", type=int, default=1_000_000, help="option host_batch: probe rows per batch")
    ap.add_argument("--pageable", action="store_true", help="input columns in pageable memory (staged through the context's pinned buffers)")
    ap.add_argument("--ctx-option", action="append", default=[])
    a = ap.parse_args()
    import numpy as np
    import hash_join_codes_knl_amd as H
    from hash_join_codes_knl_amd import api
    hj = H.HjGpu(0)
    print("library %s, library hash %s" % (os.path.basename(os.environ.get("HJGPU_LIBRARY", "libhjgpu.so")), H.library_hash()), flush=True)
    hj.set_option("host_batch", str(a.batch))
    for o in a.ctx_option:
        n, v = o.split("=")
        hj.set_option(n, v)
    fi, fo = 0x2545F491, 0x9E3779B1
    d = [hj.column(n) for n in (a.inner, a.inner, a.outer, a.outer)]
    hj.generate(1, a.inner, a.outer, 0, a.outer, fi, fo, *d)
    sums = hj.column_sums(d[2], a.outer, fo, fi)
    want = (a.outer, sums[0], sums[1], sums[2])
    if a.pageable:
        host = [c.download() for c in d]
    else:
        pinned = [hj.host_column(n) for n in (a.inner, a.inner, a.outer, a.outer)]
        for dst, src in zip(pinned, d):
            hj.lib.hjgpu_memcpy_d2h(hj.handle, dst.ptr, src.ptr, 4 * len(dst.array))
        host = [p.array for p in pinned]
    for c in d:
        c.free()
    cap = a.outer + (1 << 20)
    rows = [[hj.host_column(cap) for _ in range(3)] for _ in range(2)]          # two sets: step s is summed while step s + 1 runs
    algo = {"npj": 0, "phj": 1, "cpra": 2}[a.algo]
    pool = ThreadPoolExecutor(3)
    M = (1 << 64) - 1

    def check(step, count, cols):
        futs = [pool.submit(lambda c=c: int(c.array[:count].sum(dtype=np.uint64))) for c in cols]
        return step, count, futs

    bad, pending = 0, None

    def settle(p):
        nonlocal bad
        step, count, futs = p
        got = (count,) + tuple(f.result() & M for f in futs)
        if got != want:
            bad += 1
            if bad <= 8:
                print("step %d WRONG ROWS: count %+d, column sums %s" % (step, got[0] - want[0], ["%+d" % ((x - y + (1 << 63)) % (1 << 64) - (1 << 63)) for x, y in zip(got[1:], want[1:])]), flush=True)

    st = api.Stats()
    for s in range(a.steps):
        if s and s % 2000 == 0:
            print("... %d steps, %d wrong so far" % (s, bad), flush=True)
        cols = rows[s & 1]
        hr = api.HostRows(cols[0].ptr, cols[1].ptr, cols[2].ptr, cap)
        r = api.Result()
        hj._check(hj.lib.hjgpu_join_host_rows(hj.handle, algo, host[0].ctypes.data, host[1].ctypes.data, a.inner, host[2].ctypes.data, host[3].ctypes.data, a.outer,
                                              None, None, C.byref(hr), C.byref(r), C.byref(st)))
        if r.as_tuple() != want:
            bad += 1
            if bad <= 8:
                print("step %d WRONG AGGREGATES: count %+d" % (s, r.count - want[0]), flush=True)
        if pending:
            settle(pending)
        pending = check(s, int(r.count), cols)
    if pending:
        settle(pending)
    sd = st.as_dict()
    print("join_host_rows %s, %d x %d in batches of %d (%d batches), %s columns: %d of %d steps wrong (rows summed on the host every step); last call: device %.2f ms, upload %.1f ms"
          % (a.algo, a.inner, a.outer, a.batch, sd["batches"], "pageable" if a.pageable else "page-locked", bad, a.steps, sd["ms_total"], sd["ms_upload"]), flush=True)
    hj.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
