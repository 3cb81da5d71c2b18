#!/usr/bin/env python3
"""Enqueue-only GROUPED joins (hjgpu_phj_async / hjgpu_cpra_async with a third partitioning pass, options group_from / group_inner /
group_always) on one stream while a second context keeps another stream busy with joins of its own; EVERY step of both is checked
against the analytic aggregates.  The grouped join is planned on the device (no host thread, no wait in hardware): the call returns
at once and the caller's stream simply carries the whole plan.
usage: python tools/stress_async_grouped.py [--steps 200 --algo phj|cpra --inner N --outer N --group-inner N --neighbour 1 --rows]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--algo", default="phj", choices=["phj", "cpra"])
    ap.add_argument("--inner", type=int, default=16_000_000)
    ap.add_argument("--outer", type=int, default=64_000_000)
    ap.add_argument("--group-inner", type=int, default=4_000_000)
    ap.add_argument("--neighbour", type=int, default=1, help="0: nothing runs beside the grouped joins")
    ap.add_argument("--depth", type=int, default=2, help="grouped joins enqueued back to back before the host looks at a result")
    a = ap.parse_args()
    import numpy as np
    import torch
    torch.cuda.init()
    import hash_join_codes_knl_amd as H
    A, B = H.HjGpu(0), H.HjGpu(0)
    print("library %s, library hash %s" % (os.path.basename(os.environ.get("HJGPU_LIBRARY", "libhjgpu.so")), H.library_hash()), flush=True)
    for n, v in (("group_from", "1000"), ("group_always", "1"), ("group_inner", str(a.group_inner))):
        A.set_option(n, v)
    fi, fo = 0x2545F491, 0x9E3779B1
    M = (1 << 64) - 1
    ca = [A.column(n) for n in (a.inner, a.inner, a.outer, a.outer)]
    A.generate(1, a.inner, a.outer, 0, a.outer, fi, fo, *ca)
    sa = A.column_sums(ca[2], a.outer, fo, fi)
    want_a = [a.outer, sa[0], sa[1], sa[2]]
    nb_inner, nb_outer = 8_000_000, 48_000_000
    cb = [B.column(n) for n in (nb_inner, nb_inner, nb_outer, nb_outer)]
    B.generate(7, nb_inner, nb_outer, 0, nb_outer, fi, fo, *cb)
    sb = B.column_sums(cb[2], nb_outer, fo, fi)
    want_b = [nb_outer, sb[0], sb[1], sb[2]]
    dev = torch.device("cuda", 0)
    res_a = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(a.depth)]
    res_b = torch.zeros(4, dtype=torch.int64, device=dev)
    st_a, st_b = torch.cuda.Stream(), torch.cuda.Stream()
    fn = A.phj_async if a.algo == "phj" else A.cpra_async
    bad_a = bad_b = 0
    for s in range(a.steps):
        if s and s % 2000 == 0:
            print("... %d steps, %d grouped joins and %d neighbour joins wrong so far" % (s, bad_a, bad_b), flush=True)
        if a.neighbour:
            for _ in range(2):
                B.phj_async(cb[0], cb[1], nb_inner, cb[2], cb[3], nb_outer, None, res_b.data_ptr(), st_b.cuda_stream)
        for k in range(a.depth):
            fn(ca[0], ca[1], a.inner, ca[2], ca[3], a.outer, None, res_a[k].data_ptr(), st_a.cuda_stream)
        st_a.synchronize()
        A.get_async_status(st_a.cuda_stream)
        for k in range(a.depth):
            if [int(x) & M for x in res_a[k].tolist()] != want_a:
                bad_a += 1
                if bad_a <= 8:
                    print("step %d join %d WRONG: count %+d" % (s, k, (int(res_a[k][0]) & M) - want_a[0]), flush=True)
        if a.neighbour:
            st_b.synchronize()
            if [int(x) & M for x in res_b.tolist()] != want_b:
                bad_b += 1
    groups = A.stats()["groups"]
    print("%s_async grouped (%d groups), %d x %d, %d joins per step back to back, neighbour %d: %d of %d grouped joins wrong, %d of %d neighbour steps wrong"
          % (a.algo, groups, a.inner, a.outer, a.depth, a.neighbour, bad_a, a.steps * a.depth, bad_b, a.steps if a.neighbour else 0), flush=True)
    return 1 if (bad_a or bad_b) else 0


if __name__ == "__main__":
    sys.exit(main())
