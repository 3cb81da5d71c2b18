"""Diagnostic: hjgpu_phj_async on a non-default stream, 20 joins enqueued back to back without host syncs."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
import hash_join_codes_knl_amd as H
hj = H.HjGpu(0)
for inner, outer in ((1000, 1_000_000), (100_000, 10_000_000)):
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
    want = [outer, sums[0], sums[1], sums[2]]
    hj.reserve(inner, outer)
    d_res = torch.zeros(4, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    print("stream handle", hex(s.cuda_stream), flush=True)
    for i in range(3):
        hj.phj_async(ik, iv, inner, ok, ov, outer, None, d_res.data_ptr(), s.cuda_stream)
        s.synchronize()
        print("warm-up", i, [int(x) & ((1 << 64) - 1) for x in d_res.tolist()] == want, flush=True)
    t0 = time.perf_counter()
    for i in range(20):
        hj.phj_async(ik, iv, inner, ok, ov, outer, None, d_res.data_ptr(), s.cuda_stream)
    s.synchronize()
    print("20 back to back: %.3f ms each," % ((time.perf_counter() - t0) * 50),
          [int(x) & ((1 << 64) - 1) for x in d_res.tolist()] == want, flush=True)
    for c in (ik, iv, ok, ov):
        c.free()
print("done", flush=True)
