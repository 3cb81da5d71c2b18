#!/usr/bin/env python3
"""Repeats a one-GPU, ONE-STREAM join and checks every step against the analytic aggregates: does a library variant
(tools/build_variant.py, HJGPU_LIBRARY) give wrong results without any other stream next to it?
usage: python tools/stress_single.py [--algo phj|cpra --steps 40 --inner N --outer N]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="phj")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    ap.add_argument("--solo", action="store_true", help="option solo: partial-line stores plain (a process that runs nothing else on the device)")
    ap.add_argument("--ctx-option", action="append", default=[], help="hjgpu_set_option name=value (e.g. group_from=1000 group_always=1 group_inner=8000000)")
    ap.add_argument("--enqueue-only", action="store_true", help="the *_async form, the result read from device memory after hjgpu_synchronize")
    a = ap.parse_args()
    import hash_join_codes_knl_amd as H
    from hash_join_codes_knl_amd import api
    lib = api.load_library()
    has_dbg = hasattr(lib, "hjgpu_debug_scratch")
    if has_dbg:
        lib.hjgpu_debug_scratch.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
        lib.hjgpu_debug_scratch(None, 1)
    hj = H.HjGpu(0)
    if a.solo:
        hj.set_option("solo", "1")           # the form bench.py's headline runs in: partial-line stores plain
    for o in a.ctx_option:
        n, v = o.split("=")
        hj.set_option(n, v)
    import numpy as np
    d_res = hj.column(4, np.uint64) if a.enqueue_only else None
    ik, iv, ok, ov = hj.column(a.inner), hj.column(a.inner), hj.column(a.outer), hj.column(a.outer)
    hj.generate(1, a.inner, a.outer, 0, a.outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, a.outer, 0x9E3779B1, 0x2545F491)
    want = (a.outer, sums[0], sums[1], sums[2])
    bad = 0
    for s in range(a.steps):
        if s and s % 2000 == 0:
            print("... %d steps, %d wrong so far" % (s, bad), flush=True)
        if a.enqueue_only:
            getattr(hj, a.algo + "_async")(ik, iv, a.inner, ok, ov, a.outer, None, d_res)
            hj.synchronize()
            hj.get_async_status()
            got = tuple(int(x) for x in d_res.download())
        else:
            got = getattr(hj, a.algo)(ik, iv, a.inner, ok, ov, a.outer)
        if tuple(got) != want:
            bad += 1
            if bad <= 6:
                print("step %d WRONG: count %+d" % (s, got[0] - want[0]), flush=True)
    st = hj.stats()
    print("%s%s one stream%s, library %s (%s): %d of %d steps wrong; last step %.2f ms (pass 1 %.2f, pass 2 %.2f, join %.2f)"
          % (a.algo, "_async" if a.enqueue_only else "", (", options " + " ".join(a.ctx_option) + ", %d groups" % st["groups"]) if a.ctx_option else "", os.path.basename(os.environ.get("HJGPU_LIBRARY", "libhjgpu.so")), "library hash " + H.library_hash(), bad, a.steps,
             st["ms_total"], st["ms_scatter1"], st["ms_scatter2"], st["ms_join"]), flush=True)
    if has_dbg:
        d = (ctypes.c_uint64 * 40)()
        lib.hjgpu_debug_scratch(d, 0)
        print("private segment: %d values re-read and compared in the kernel, %d MISMATCHES" % (d[1], d[0]), flush=True)
    hj.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
