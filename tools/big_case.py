"""BASELINE config 5's per-GPU shape on one GPU: |R| = 128 M build tuples (1 G / 8 GPUs) and a probe
side of more than 2^31 tuples (16 G / 8 = 2 G per GPU): 64-bit offsets everywhere, fan-out at the
library's limits.  Prints the aggregates against the column sums of S and the phase times."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
torch.cuda.init()
import hash_join_codes_knl_amd as H

inner = int(sys.argv[1]) if len(sys.argv) > 1 else 128_000_000
outer = int(sys.argv[2]) if len(sys.argv) > 2 else 2_200_000_000
fi, fo = 0x2545F491, 0x9E3779B1
with H.HjGpu(0) as hj:
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(3, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, fo, fi)
    want = (outer, sums[0], sums[1], sums[2])
    hj.reserve(inner, outer)
    for name, fn, prm in (("phj", hj.phj, None), ("cpra", hj.cpra, H.PhjParams(chunks=8)), ("npj", hj.npj, None)):
        for rep in range(2):
            got = fn(ik, iv, inner, ok, ov, outer, prm) if prm is not None else fn(ik, iv, inner, ok, ov, outer)
        st = hj.stats()
        print(name, "ok" if got == want else "MISMATCH %r vs %r" % (got, want),
              {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if v}, flush=True)
