"""Multi-PROCESS CPU tests over the gloo backend, world_size 2 and 3: (a) bench.py's own control plane (id broadcast, max over
ranks, uint64 sums, the gather of every rank's statistics: what `bench.py --gpus N` does under torchrun beside the C++ joins),
and (b) the ALGORITHM of the two exchanges - ownership, split sizes, all-to-all-v, reductions - stated a third time
(tests/torch_orchestration.py, round 1's orchestration over torch.distributed with the oracle as the data path).
(b) is NOT the product's code: the product's C++ orchestration (csrc/hjgpu_multi.hip) is tested directly, on the CPU, by
tests/test_pipeline_ordering.py (round 5) and on the GPU by tests/test_gpu_multi.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from oracle_ops import OracleOps, PreparedOracleOps
    import torch_orchestration as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ik, iv, ok, ov = O.generate(40_000, 9_000, seed=5)
        want = O.join_definition(ik, iv, ok, ov)
        ops = PreparedOracleOps(O, torch) if mode == "cpra_prepared" else OracleOps(O, torch)
        as_t = lambda a: torch.from_numpy(a.view(np.int32).copy())
        sb = D.shard_bounds(len(ok), world)[rank]
        if mode == "phj":
            rk = as_t(ik) if rank == 0 else torch.zeros(len(ik), dtype=torch.int32)
            rv = as_t(iv) if rank == 0 else torch.zeros(len(iv), dtype=torch.int32)
            got = D.phj_replicated_build(dist, torch, ops, rk, rv, as_t(ok[sb[0]:sb[1]]), as_t(ov[sb[0]:sb[1]]))
        else:
            rb = D.shard_bounds(len(ik), world)[rank]
            # mode "cpra_rounds": force the chunked exchange (several rounds through staging buffers);
            # "cpra": probe side in 4 slices whose transfers are asynchronous; "cpra_tiny_slices": more
            # slices than 16-tuple units in a shard, so most slices are empty
            got = D.cpra_copartitioned(dist, torch, ops, as_t(ik[rb[0]:rb[1]]), as_t(iv[rb[0]:rb[1]]),
                                       as_t(ok[sb[0]:sb[1]]), as_t(ov[sb[0]:sb[1]]),
                                       max_elems=(1500 if mode == "cpra_rounds" else D.MAX_MESSAGE_ELEMS),
                                       slices={"cpra": 4, "cpra_rounds": 2, "cpra_one_slice": 1, "cpra_tiny_slices": 300,
                                               "cpra_prepared": 5}[mode])
            if mode == "cpra_prepared":          # the received build side was prepared once, every slice probed it
                assert (ops.builds, ops.probes) == (1, 5)
        q.put((rank, got == want, got, want))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["phj", "cpra", "cpra_rounds", "cpra_one_slice", "cpra_tiny_slices", "cpra_prepared"])
def test_multi_process_join(world, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, got, want in results:
        assert ok, (rank, got, want)


def _replicate_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import torch_orchestration as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        want = torch.arange(n, dtype=torch.int32) * 7 - 3
        t = want.clone() if rank == 1 % world else torch.zeros(n, dtype=torch.int32)
        D.replicate(dist, torch, t, src=1 % world)
        q.put((rank, bool(torch.equal(t, want))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 100_003), (3, 50_000), (3, 17)])
def test_replicate_scatter_allgather(world, n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replicate_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


def test_shard_bounds_and_ownership():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_orchestration as D
    b = D.shard_bounds(1000, 3)
    assert b == [(0, 320), (320, 640), (640, 1000)]           # npj.cpp:516-529 with alignment 16
    assert [D.owner_of_partition(p, 10, 4) for p in range(10)] == [0, 0, 1, 1, 2, 2, 3, 3, 3, 3]


def _control_plane_worker(rank, world, port, q):
    """bench.py's control plane under torchrun, on the gloo group it really uses: the ncclUniqueId drawn on rank 0
    reaches every rank unchanged (the data plane - hjgpu_comm_create_rank - is replaced by a recorder: no GPU here),
    the step time is the maximum over the ranks, the expected aggregates add up with uint64 wrap-around."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class FakeComm:
            @staticmethod
            def new_id():
                return bytes(range(128))               # drawn on rank 0 only

            @staticmethod
            def rank(device, nranks, r, comm_id):
                return ("joined", device, nranks, r, comm_id)

        class FakeH:
            HjComm = FakeComm
        joined = bench.connect_ranks(dist, FakeH, 10 + rank, rank, world)
        slowest = bench.max_over_ranks(dist, torch, 0.5 + rank)
        sums = bench.sum_over_ranks(dist, torch, [1_000, (1 << 64) - 5, 7, 1 << 63])
        # round 3: every rank's exchange statistics reach every rank (min / max / mean over ALL ranks in the result line)
        every = bench.gather_objects(dist, {"rank": rank, "ms": 1.0 + rank})
        assert [e["rank"] for e in every] == list(range(world))
        assert bench.spread([e["ms"] for e in every]) == {"min": 1.0, "max": float(world), "mean": round((world + 1) / 2, 4)}
        q.put((rank, joined, slowest, sums))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_control_plane_over_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_control_plane_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mask = (1 << 64) - 1
    for rank, joined, slowest, sums in results:
        assert joined == ("joined", 10 + rank, world, rank, bytes(range(128)))
        assert slowest == 0.5 + world - 1
        assert sums == [1_000 * world, ((1 << 64) - 5) * world & mask, 7 * world, (1 << 63) * world & mask]
