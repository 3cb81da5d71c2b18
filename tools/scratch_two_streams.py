#!/usr/bin/env python3
"""Two INDEPENDENT library calls on two streams, no slice pipeline, no communicator: context A partitions a relation
(hjgpu_partition_packed_async = K4 + K5 + K6 pass 1, the exchange-level pass of the multi-GPU CPRA) over and over on its
stream while context B runs whole PHJ joins on another stream.  Every partition output is checked on A's own stream:
sum of keys and payloads = the input's, every tuple in the partition its key hashes to.  With the product library both
stay right; with a variant whose pass-1 kernel carries a private segment (tools/build_variant.py scratch_exp9
-DHJ_SCRATCH_EXPERIMENT=9: one private word, written once, never read) the partition output loses stores - but only
while B's kernels run next to it.
usage: HJGPU_LIBRARY=<variant.so> python tools/scratch_two_streams.py [--steps 40 --n 125000000 --fanout 192 --neighbour 1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--n", type=int, default=125_000_000, help="tuples partitioned per step (one slice of the pipeline)")
    ap.add_argument("--fanout", type=int, default=192)
    ap.add_argument("--neighbour", type=int, default=1, help="0: nothing runs next to the partitioning")
    ap.add_argument("--unpacked", action="store_true", help="hjgpu_partition_async (separate key / payload columns out: another stream-out path of K6) instead of the packed operator")
    ap.add_argument("--option", action="append", default=[], help="name=value, hjgpu_set_option on the PARTITIONING context (e.g. scatter_cfg=512,4,1)")
    ap.add_argument("--recheck", action="store_true", help="on a wrong output: synchronise the device, copy the output to the host with hipMemcpy "
                    "(SDMA: not through any XCD's L2) and count there, then count again with a fresh kernel - is the store LOST in memory or was the first read STALE?")
    ap.add_argument("--quiet-after", type=int, default=5, help="wrong outputs described in full")
    a = ap.parse_args()
    import numpy as np
    import torch
    import hash_join_codes_knl_amd as H
    dev = torch.device("cuda", 0)
    A, B = H.HjGpu(0), H.HjGpu(0)
    for o in a.option:
        name, _, value = o.partition("=")
        A.set_option(name, value)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    inner, outer = 64_000_000, 1_000_000_000
    ik, iv, ok, ov = B.column(inner), B.column(inner), B.column(outer), B.column(outer)
    B.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = B.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
    want_join = [outer, sums[0], sums[1], sums[2]]
    d_res = torch.zeros(4, dtype=torch.int64, device=dev)
    n, F, factor = a.n, a.fanout, 0x2C1B3C6D
    out = torch.zeros(n + 64, dtype=torch.int64, device=dev)
    off = torch.zeros(F + 1, dtype=torch.int64, device=dev)
    # the slice that is partitioned: rows [0, n) of the probe side; its sums from torch
    # torch views of the library-owned columns
    def as_tensor(col, count):
        class _Holder:                                           # __cuda_array_interface__ over a raw device pointer
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i4", "data": (int(col.ptr), False), "version": 2}
        return torch.as_tensor(h, device=dev)
    tk, tv = as_tensor(ok, n), as_tensor(ov, n)
    want_k = int(tk.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())
    want_v = int(tv.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())
    torch.cuda.synchronize()
    bad_part, bad_join, lost_slots = 0, 0, []
    recheck_log = []
    for s in range(a.steps):
        if a.neighbour:
            for _ in range(3):                                    # ~25 ms of join kernels next to ~1 ms of partitioning
                B.phj_async(ik, iv, inner, ok, ov, outer, None, d_res.data_ptr(), sB.cuda_stream)
        with torch.cuda.stream(sA):
            for _ in range(4):                                    # several partition calls inside the neighbour's run
                out.zero_()                                       # a slot that is not written stays 0
                if a.unpacked:
                    A.partition_async(ok, ov, n, factor, F, out.data_ptr(), out.data_ptr() + 4 * (n + 32), off.data_ptr(), sA.cuda_stream)
                else:
                    A.partition_packed_async(ok, ov, n, factor, F, out.data_ptr(), off.data_ptr(), sA.cuda_stream)
            if a.unpacked:
                o32 = out.view(torch.int32)
                kk, vv = o32[:n].to(torch.int64).bitwise_and(0xFFFFFFFF), o32[n + 32:2 * n + 32].to(torch.int64).bitwise_and(0xFFFFFFFF)
                got_k, got_v = int(kk.sum().item()), int(vv.sum().item())
                t = kk + (vv << 32)
            else:
                t = out[:n]
                got_k = int(t.bitwise_and(0xFFFFFFFF).sum().item())
                got_v = int((t >> 32).bitwise_and(0xFFFFFFFF).sum().item())
            unwritten = int((t == 0).sum().item())
            where = (t == 0).nonzero().flatten()[:4096].tolist() if unwritten else []
        torch.cuda.synchronize()
        if (got_k, got_v) != (want_k, want_v) and a.recheck and not a.unpacked:
            # the device is quiet now.  (1) the buffer as the HOST sees it through hipMemcpy; (2) as a fresh kernel sees it
            host = np.empty(n, dtype=np.uint64)
            A._check(A.lib.hjgpu_memcpy_d2h(A.handle, host.ctypes.data, out.data_ptr(), n * 8))
            hk = int((host & np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64)); hv = int((host >> np.uint64(32)).sum(dtype=np.uint64))
            hz = int((host == 0).sum())
            t2 = out[:n]
            k2 = int(t2.bitwise_and(0xFFFFFFFF).sum().item()); z2 = int((t2 == 0).sum().item())
            torch.cuda.synchronize()
            M = (1 << 64) - 1
            verdict = "LOST in memory" if hz else ("memory is right: the first read was STALE" if (hk & M) == (want_k & M) else "memory differs without zero slots")
            recheck_log.append((unwritten, hz, z2))
            if bad_part < a.quiet_after:
                print("step %d recheck: first kernel read saw %d unwritten slots; device quiet: hipMemcpy to the host sees %d unwritten slots (key sum %+d), "
                      "a fresh kernel sees %d  ->  the stores are %s" % (s, unwritten, hz, hk - want_k, z2, verdict), flush=True)
        if (got_k, got_v) != (want_k, want_v):
            bad_part += 1
            if bad_part <= a.quiet_after:
                print("step %d: partition output WRONG: %d of %d slots never written; key sum %+d, payload sum %+d"
                      % (s, unwritten, n, got_k - want_k, got_v - want_v), flush=True)
                # the shape of the loss: maximal runs of consecutive unwritten slots (first row, length)
                runs, start, prev = [], None, None
                for x in where:
                    if start is None: start, prev = x, x
                    elif x == prev + 1: prev = x
                    else: runs.append((start, prev - start + 1)); start, prev = x, x
                if start is not None: runs.append((start, prev - start + 1))
                lens = {}
                for _, ln in runs: lens[ln] = lens.get(ln, 0) + 1
                print("    runs of unwritten slots (length: how many) %s; first runs (row, row %% 16, length): %s"
                      % (sorted(lens.items()), [(r, r % 16, ln) for r, ln in runs[:12]]), flush=True)
        if a.neighbour and [int(x) & ((1 << 64) - 1) for x in d_res.tolist()] != want_join:
            bad_join += 1
    if a.recheck:
        same = sum(1 for u, h, z in recheck_log if u == h == z)
        print("recheck: %d wrong outputs looked at again with the device quiet; in %d of them the host copy (hipMemcpy) and a fresh kernel count exactly the "
              "unwritten slots the first read saw; host sees none (a stale first read) in %d" % (len(recheck_log), same, sum(1 for u, h, z in recheck_log if h == 0)), flush=True)
    print("library %s (%s), options %s, neighbour %d: %d of %d partition outputs wrong, %d of %d neighbour joins wrong"
          % (os.path.basename(os.environ.get("HJGPU_LIBRARY", "libhjgpu.so")), H.kernel_hash(), a.option, a.neighbour, bad_part, a.steps,
             bad_join, a.steps if a.neighbour else 0), flush=True)


if __name__ == "__main__":
    main()
