import os, sys
sys.path.insert(0, os.getcwd())
import hash_join_codes_knl_amd as H
hj = H.HjGpu(0)
inner, outer = 64_000_000, 1_000_000_000
ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
for i in range(2): hj.phj(ik, iv, inner, ok, ov, outer)
os.environ["HJGPU_SCATTER_PROF"] = "1"
hj.phj(ik, iv, inner, ok, ov, outer)
os.environ["HJGPU_SCATTER_CFG"] = "1024,4,0"
hj.phj(ik, iv, inner, ok, ov, outer)
