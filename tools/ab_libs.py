#!/usr/bin/env python3
"""A/B timing of two (or more) builds of libhjgpu.so INSIDE ONE PROCESS, interleaved, same relations:
usage: python tools/ab_libs.py <a.so> <b.so> ... [--rounds R --reps K --inner N --outer N --algo phj]
(variants: tools/build_variant.py).  Every join is checked against the analytic aggregates.
--sequential: K6's time depends on WHICH allocation holds its output (placement), so contexts that live
side by side compare their workspaces' luck as much as their kernels.  In this mode only one measured context
exists at a time: it is created (placement=1: the first allocation), timed and destroyed, round after round - the
allocator hands every context the blocks the previous one freed, so all builds write into the same memory."""
import argparse
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--algo", default="phj")
    ap.add_argument("--no-check", action="store_true", help="timing experiments with deliberately wrong variants")
    ap.add_argument("--sequential", action="store_true", help="one measured context at a time, same allocations for all")
    ap.add_argument("--rows", action="store_true", help="materialise the result (three columns shared by all builds)")
    a = ap.parse_args()
    import hash_join_codes_knl_amd as H
    from hash_join_codes_knl_amd import api
    def make(path):
        os.environ["HJGPU_LIBRARY"] = os.path.abspath(path)
        api._lib = None                      # the next context binds (and keeps) this build
        return H.HjGpu(0)
    ctxs = [make(a.libs[0])] if a.sequential else [make(path) for path in a.libs]
    hj = ctxs[0]                             # owns the relations
    ik, iv, ok, ov = hj.column(a.inner), hj.column(a.inner), hj.column(a.outer), hj.column(a.outer)
    hj.generate(1, a.inner, a.outer, 0, a.outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, a.outer, 0x9E3779B1, 0x2545F491)
    want = (a.outer, sums[0], sums[1], sums[2])
    phases = ["ms_total", "ms_histogram", "ms_plan", "ms_scatter1", "ms_scatter2", "ms_join", "ms_build", "ms_close_gaps"]
    kw = {}
    if a.rows:
        block = 16384
        cap = ((a.outer + block - 1) // block + 4096 + 8) * block
        kw["out"] = (hj.column(cap), hj.column(cap), hj.column(cap), cap, block)
    data = {p: {ph: [] for ph in phases} for p in a.libs}
    for rnd in range(a.rounds if a.sequential else 0):
        for path in a.libs:
            c = make(path)
            c.set_option("placement", "1")
            for rep in range(a.reps + 1):
                got = getattr(c, a.algo)(ik, iv, a.inner, ok, ov, a.outer, **kw)
                assert a.no_check or got == want, (path, got, want)
                st = c.stats()
                if rep > 0:                  # the first join allocates the workspace
                    for ph in phases:
                        data[path][ph].append(st[ph])
            c.close()
    for rnd in range(0 if a.sequential else a.rounds):
        for path, c in zip(a.libs, ctxs):
            for _ in range(a.reps):
                got = getattr(c, a.algo)(ik, iv, a.inner, ok, ov, a.outer, **kw)
                assert a.no_check or got == want, (path, got, want)
                st = c.stats()
                if rnd > 0 or a.rounds == 1:
                    for ph in phases:
                        data[path][ph].append(st[ph])
    for path in a.libs:
        print("%-28s" % os.path.basename(path), " ".join("%s %.3f/%.3f |" % (ph[3:], statistics.median(x), min(x))
                                                          for ph, x in data[path].items() if max(x) > 0), flush=True)


if __name__ == "__main__":
    main()
