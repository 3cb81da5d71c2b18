#!/usr/bin/env python3
"""NPJ on small relations (duplicate-heavy build sides, so every walk leaves its first line) at
several load factors, against the oracle's join definition: the quick check used while the
line-hashed cooperative probe was brought up.  Run under `timeout`: a wrong walk may not end."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hash_join_codes_knl_amd as H
from oracle import oracle as O
hj = H.HjGpu(0)
for (inner, outer) in ((2048, 16384), (8192, 512), (100000, 50000)):
    ik, iv, ok, ov = O.generate(inner, outer, seed=3)
    want = O.join_definition(ik, iv, ok, ov)
    rk, rv, sk, sv = hj.column(ik), hj.column(iv), hj.column(ok), hj.column(ov)
    for load in (0.25, 0.5, 0.75, 0.9):
        print("inner", inner, "outer", outer, "load", load, flush=True)
        got = hj.npj(rk, rv, len(ik), sk, sv, len(ok), H.NpjParams(load=load))
        print("   ", got == want, got, want, flush=True)
