#!/usr/bin/env python3
"""Dense primary keys (1..|R|, the common real-world build side) instead of the generator's random unique keys:
two multiplicative hashes of consecutive integers form a lattice, so partition sizes and cuckoo-table occupancy
are not those of random keys.  Prints the phase times of PHJ / NPJ / CPRA for both kinds of relation and checks the
aggregates against torch.   usage: python tools/sequential_keys.py [inner] [outer]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hash_join_codes_knl_amd as H

inner = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
outer = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
dev = torch.device("cuda:0")
hj = H.HjGpu(0)
M = (1 << 32) - 1


def u32(t):
    return t.to(torch.int64).bitwise_and(M)


def run(name, ik, iv, ok, ov):
    # every probe key matches exactly one build key: the expected aggregates are sums over the probe side
    want = (outer, int(u32(ok).sum().item()), int(u32(ov).sum().item()), None)
    for algo in ("phj", "npj", "cpra"):
        best = None
        for _ in range(3):
            got = getattr(hj, algo)(ik.data_ptr(), iv.data_ptr(), inner, ok.data_ptr(), ov.data_ptr(), outer)
            st = hj.stats()
            if best is None or st["ms_total"] < best["ms_total"]:
                best = st
        assert got[:3] == want[:3], (name, algo, got, want)
        print("%-10s %-5s total %.2f ms | hist %.2f plan %.2f scatter1 %.2f scatter2 %.2f join %.2f build %.2f | fan-out %dx%d"
              % (name, algo, best["ms_total"], best["ms_histogram"], best["ms_plan"], best["ms_scatter1"], best["ms_scatter2"],
                 best["ms_join"], best["ms_build"], best.get("fanout1", 0), best.get("fanout2", 0)), flush=True)


g = torch.Generator(device=dev)
g.manual_seed(1)
# dense keys 1..inner, payload = key * 7; probe side = uniform picks
ik = torch.arange(1, inner + 1, dtype=torch.int64, device=dev).to(torch.int32)
iv = (ik.to(torch.int64) * 7).bitwise_and(M).to(torch.int32)
ok = torch.randint(1, inner + 1, (outer,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
ov = (ok.to(torch.int64) * 3).bitwise_and(M).to(torch.int32)
run("dense", ik, iv, ok, ov)
del ik, iv, ok, ov
# the generator's relations (random unique keys) in the same process
ik, iv, ok, ov = (torch.empty(n, dtype=torch.int32, device=dev) for n in (inner, inner, outer, outer))
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik.data_ptr(), iv.data_ptr(), ok.data_ptr(), ov.data_ptr())
run("random", ik, iv, ok, ov)
