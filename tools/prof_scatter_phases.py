#!/usr/bin/env python3
"""Where a scatter workgroup's time goes (option scatter_prof: s_memtime stamps between the
barriers of K6, printed by the library).  usage: python tools/prof_scatter_phases.py [zipf]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hash_join_codes_knl_amd as H

zipf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
hj = H.HjGpu(0)
inner, outer = 64_000_000, 1_000_000_000
ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
hj.generate_zipf(1, inner, outer, 0, inner, 0, outer, 0x2545F491, 0x9E3779B1, zipf, ik, iv, ok, ov)
for i in range(2):
    hj.phj(ik, iv, inner, ok, ov, outer)
hj.set_option("scatter_prof", 1)
hj.phj(ik, iv, inner, ok, ov, outer)
