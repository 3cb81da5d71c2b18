#!/usr/bin/env python3
"""PHJ step time when another kernel holds some compute units (what an RCCL transfer of the build
side does on the multi-GPU path): a hog of N workgroups (one CU each) is launched on a side stream
for the first `us` microseconds of every join.  usage: python tools/contention_check.py [N] [us]
(builds tools/cu_hog.hip into hash_join_codes_knl_amd/lib/libcuhog.so on first use)"""
import ctypes
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    torch.cuda.init()
    import hash_join_codes_knl_amd as H
    n_wgs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    us = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    so = os.path.join(os.path.dirname(os.path.abspath(H.__file__)), "lib", "libcuhog.so")
    if not os.path.exists(so):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "cu_hog.hip"), "-o", so])
    hog = ctypes.CDLL(so)
    hog.hog_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    hj = H.HjGpu(0)
    inner, outer = 64_000_000, 1_000_000_000
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
    want = (outer, sums[0], sums[1], sums[2])
    hj.reserve(inner, outer)
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    d_res = torch.zeros(4, dtype=torch.int64, device="cuda")
    for label, hogs in (("alone", 0), ("with %d CUs held for %d us" % (n_wgs, us), n_wgs), ("alone", 0)):
        tot = []
        for i in range(6):
            torch.cuda.synchronize()
            if hogs:
                assert hog.hog_launch(side.cuda_stream, hogs, us) == 0
            hj.phj_async(ik, iv, inner, ok, ov, outer, None, d_res.data_ptr(), main.cuda_stream)
            torch.cuda.synchronize()
            assert tuple(int(x) & (2**64 - 1) for x in d_res.tolist()) == want
            if i:
                tot.append(hj.stats()["ms_total"])
        print("%-36s join %.2f ms (min %.2f)" % (label, statistics.median(tot), min(tot)), flush=True)


if __name__ == "__main__":
    main()
