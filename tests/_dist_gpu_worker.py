"""Worker of test_gpu_distributed.py: the multi-GPU orchestration (distributed.py) driving the C-ABI through
GpuOps over RCCL ("nccl" backend) at world size 1 — the only world a one-GPU box offers; world 2 and 3 run over
gloo with the oracle injected (test_distributed_gloo.py).  Prints one line per case and "ALL OK"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    torch.cuda.init()
    import torch.distributed as dist
    import hash_join_codes_knl_amd as H
    import torch_orchestration as D
    from helpers import numpy_join
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ok = True
    try:
        with H.HjGpu(0) as hj:
            rng = np.random.default_rng(8)
            base = np.unique(rng.integers(1, 2**32, size=300_000, dtype=np.uint64).astype(np.uint32))
            ik = np.concatenate([base, base[:20_000]])
            iv = rng.integers(0, 2**32, size=len(ik), dtype=np.uint64).astype(np.uint32)
            okeys = np.where(rng.random(2_000_000) < 0.8, base[rng.integers(0, len(base), size=2_000_000)],
                             rng.integers(1, 2**32, size=2_000_000, dtype=np.uint64).astype(np.uint32)).astype(np.uint32)
            ov = rng.integers(0, 2**32, size=len(okeys), dtype=np.uint64).astype(np.uint32)
            want = numpy_join(ik, iv, okeys, ov)
            t = lambda a: torch.from_numpy(a.view(np.int32).copy()).to(dev)
            ops = D.GpuOps(hj, torch, "phj", None)
            hj2 = H.HjGpu(0)
            ops2 = D.GpuOps(hj, torch, "phj", None, partition_ctx=hj2)     # prepared build side across the slices
            cases = [("phj_replicated_build", lambda: D.phj_replicated_build(dist, torch, ops, t(ik), t(iv), t(okeys), t(ov)))]
            for slices, max_elems in ((1, D.MAX_MESSAGE_ELEMS), (4, D.MAX_MESSAGE_ELEMS), (3, 100_000), (5000, D.MAX_MESSAGE_ELEMS)):
                cases.append(("cpra_copartitioned slices=%d max_elems=%d" % (slices, max_elems),
                              lambda s=slices, m=max_elems: D.cpra_copartitioned(dist, torch, ops, t(ik), t(iv), t(okeys), t(ov),
                                                                               max_elems=m, slices=s)))
            for slices, max_elems in ((4, D.MAX_MESSAGE_ELEMS), (7, 100_000)):
                cases.append(("cpra_copartitioned prepared build, slices=%d max_elems=%d" % (slices, max_elems),
                              lambda s=slices, m=max_elems: D.cpra_copartitioned(dist, torch, ops2, t(ik), t(iv), t(okeys), t(ov),
                                                                               max_elems=m, slices=s)))
            for name, fn in cases:
                got = tuple(fn())
                print(name, "OK" if got == want else "MISMATCH %r != %r" % (got, want), flush=True)
                ok = ok and got == want
            # the slice joins were logged with their sizes (bench.py's accounting)
            ok = ok and len(ops.join_log) >= 1 + 1 + 4 + 3
            # one build + one probe per slice (a batch beyond max_outer would add pieces)
            ok = ok and [c["inner"] > 0 for c in ops2.join_log].count(True) == 2
            hj2.close()
    finally:
        dist.destroy_process_group()
    print("ALL OK" if ok else "FAILED", flush=True)


if __name__ == "__main__":
    main()
