"""Which HIP runtime(s) does a process that uses torch + libhjgpu map?  (They must be ONE:
streams and events are passed between the two.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import hash_join_codes_knl_amd as H
hj = H.HjGpu(0)
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "libhsa-runtime" in l})
print("\n".join(libs))
