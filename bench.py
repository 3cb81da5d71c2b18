#!/usr/bin/env python3
"""bench.py — PHJ |R|=64M join |S|=1G per GPU on MI355X (BASELINE.json configs[2];
with --gpus N: configs[3], build side replicated over RCCL, probe side sharded; --algo cpra --gpus N: configs[4]'s
shape, both sides chunked and co-partitioned by an all-to-all-v).

One "step" = one complete partitioned hash join over HBM-resident columns:
fused histogram (R,S) -> plan -> scatter pass 1 -> scatter pass 2 -> LDS
build+probe, aggregates (count + 3 checksums) resident in HBM at the end.
Nothing is skipped or cached between steps; the join result of every step is
checked against the analytic aggregates of the generated relations.

N > 1: one process per GPU (torchrun); the data path - exchange over RCCL, local joins, all-reduce - is the
library's C++ (hjgpu_phj_multi / hjgpu_cpra_multi, include/hjgpu.h); torch.distributed is the control plane only
(gloo: the ncclUniqueId reaches the ranks, barriers around the timed region, max over ranks of the time).

Prints ONE JSON line on rank 0 (see the driver contract in the task).  At N = 1 the line also carries, measured in
the same process: NPJ (configs[1]) and one-GPU CPRA (`secondary`), the materialising PHJ (`materialized_default`, `materialized`), the
CPU restatement of the same PHJ (`cpu_baseline`) and of configs[0] (`cpu_baseline_config0`).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)
INNER_FACTOR, OUTER_FACTOR = 0x2545F491, 0x9E3779B1
# PMC traffic of the kernels (tools/collect_traffic.py): THIS file, and only while its kernel hash is the
# running library's (hjgpu_kernel_hash) and it was taken on the workload being run
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_traffic.json")
# the other legs of the N = 1 line: PMC traffic of the same kernels measured through `tools/collect_traffic.py <args>` (the file
# records the arguments); attached under the same rule as the headline's - same kernel hash, same workload - or null with the reason
SECONDARY_TRAFFIC = {"npj": ("r06_npj_traffic.json", ["--algo", "npj"]),
                     "cpra": ("r06_cpra_traffic.json", ["--algo", "cpra"]),
                     "phj_unique": ("r06_unique_traffic.json", ["--option", "unique=1"]),
                     "materialized_default": ("r06_materialized_traffic.json", ["--materialized"])}
MASK64 = (1 << 64) - 1


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--algo", choices=["phj", "npj", "cpra"], default="phj")
    ap.add_argument("--inner", type=int, default=64_000_000, help="|R| build tuples (replicated; --algo cpra --gpus N: per GPU)")
    ap.add_argument("--outer", type=int, default=1_000_000_000, help="|S| probe tuples PER GPU")
    ap.add_argument("--zipf", type=float, default=0.0,
                    help="Zipf exponent of the probe side's repeat picks (0 = uniform, the headline workload)")
    ap.add_argument("--fanout1", type=int, default=0)
    ap.add_argument("--fanout2", type=int, default=0)
    ap.add_argument("--cpu-outer", type=int, default=1_000_000_000,
                    help="probe tuples of the CPU-baseline sample (0 = skip every CPU leg); the default is the whole per-GPU "
                         "workload: ~1 s (~18 CPU-seconds) on the 16 CPUs of a GPU box with the AVX-512 operators")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="0 = the CPUs this process may use (affinity mask capped by the cgroup CPU quota)")
    ap.add_argument("--enqueue-only", action="store_true",
                    help="N = 1: time the enqueue-only entry points (hjgpu_*_async: all K6 stores non-temporal) instead of the blocking ones")
    ap.add_argument("--solo", action="store_true", help="N = 1: the headline with option solo (the caller's promise that nothing else runs on the device: K6's partial-line "
                    "stores and the result rows plain); the default line measures the library's DEFAULT policy and reports the solo form in secondary.phj_solo")
    ap.add_argument("--no-solo", action="store_true", help="kept for compatibility: the default since round 6")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra N = 1 measurements (NPJ, one-GPU CPRA, materialising PHJ, configs[0] on the CPU)")
    ap.add_argument("--materialize", action="store_true", help="kept for compatibility: the materialising PHJ is on by default")
    ap.add_argument("--force-dist", action="store_true",
                    help="at --gpus 1: go through the multi-GPU entry points with a one-rank RCCL communicator")
    ap.add_argument("--ring-broadcast", action="store_true",
                    help="multi-GPU: replicate the build side with one ncclBroadcast instead of scatter + all-gather")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="multi-GPU: CUs the partitioning kernels leave to RCCL's kernels (-1 = the library's default: 16 with > 1 rank)")
    ap.add_argument("--comm-option", action="append", default=[], metavar="NAME=VALUE",
                    help="communicator options for measurements (hjgpu_comm_set_option), e.g. cpra_k=24, cpra_two_level=1")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="context options for measurements (hjgpu_set_option), e.g. group_from=0, placement=12")
    ap.add_argument("--exchange-slices", type=int, default=0,
                    help="multi-GPU CPRA: pieces the probe side travels in (partition / all-to-all / join overlap); "
                         "0 = the library's choice: 4, or 1 in a world of one (nothing to overlap)")
    ap.add_argument("--comm-timeout-ms", type=int, default=120_000,
                    help="multi-GPU: deadline of every host-side wait inside the library (hjgpu_comm option timeout_ms); when it "
                         "expires the communicator is aborted (ncclCommAbort) and this process exits with status 3")
    ap.add_argument("--rehearse-solo", action="store_true",
                    help="rehearsal of the N > 1 control flow on a box with ONE GPU: every rank of the torchrun world runs on "
                         "device 0 with its own one-rank RCCL communicator (no exchange between the processes: RCCL wants one "
                         "device per rank), so that everything around the data plane - gloo gathers, barriers, per-rank "
                         "statistics, the N > 1 JSON line - executes with more than one process; its numbers mean nothing")
    ap.add_argument("--configs4-inner", type=int, default=128_000_000,
                    help="N > 1 line, secondary.cpra_multi (BASELINE configs[4]: CPRA |R|=1 G x |S|=16 G on 8 GPUs): build tuples PER GPU")
    ap.add_argument("--configs4-outer", type=int, default=2_000_000_000,
                    help="secondary.cpra_multi: probe tuples PER GPU")
    ap.add_argument("--configs4-steps", type=int, default=4, help="secondary.cpra_multi: timed steps (after one warm-up step)")
    ap.add_argument("--preflight-bytes", type=int, default=256 << 20,
                    help="multi-GPU: bytes per peer of the link-bandwidth preflight before the timed region (0 = skip it)")
    return ap.parse_args(argv)


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child
    process group BEFORE anything touches the GPU, and exit with its code."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands
    a one-GPU job 16 CPUs' worth of time out of 256 online ones; 256 runnable threads would only be throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def connect_ranks(dist, H, local_rank, rank, world):
    """Control plane -> data plane: rank 0 draws the ncclUniqueId (hjgpu_comm_get_id), the gloo group carries it to
    every rank, each rank joins the library's communicator (hjgpu_comm_create_rank = ncclCommInitRank)."""
    # one node by contract: RCCL's bootstrap sockets need no interface beyond loopback (the box's hostname or its
    # outward interfaces may not resolve / route); an explicit NCCL_SOCKET_IFNAME of the caller wins
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    box = [H.HjComm.new_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return H.HjComm.rank(local_rank, world, rank, box[0])


def gather_objects(dist, obj):
    """every rank's `obj` on every rank, over the gloo control plane (None without one)"""
    if dist is None:
        return [obj]
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, obj)
    return box


def spread(values):
    values = [float(v) for v in values]
    return {"min": round(min(values), 4), "max": round(max(values), 4), "mean": round(sum(values) / len(values), 4)}


def die_of_comm_error(rank, ex, where):
    """A multi-GPU call failed (deadline expired -> communicator aborted, RCCL error, wrong preflight data): say so and
    leave with a FRESH exit - no re-exec, no teardown that could wait for a dead peer."""
    sys.stderr.write(json.dumps({"bench_error": "rank %d, %s: %s" % (rank, where, ex)}) + "\n")
    sys.stderr.flush()
    sys.stdout.flush()
    os._exit(3)


def max_over_ranks(dist, torch, seconds):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, torch, values):
    """uint64 aggregates as int64 bit patterns (wrap-around addition is the same)."""
    if dist is None:
        return [int(v) & MASK64 for v in values]
    t = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in values], dtype=torch.int64)
    dist.all_reduce(t)
    return [int(v) & MASK64 for v in t.tolist()]


def cpu_baseline(hj, args, algo, more=()):
    """The oracle's restatement of the reference's CPU algorithm ("port"), timed on
    this host's cores on a bounded sample of the same workload shape.  `more`: further algorithms joined on the SAME host columns
    (one generation, one download): the entries come back under "others" (the NPJ and CPRA legs' baselines)."""
    from oracle import oracle as O
    outer = min(args.cpu_outer, args.outer)
    inner = max(1, int(args.inner * (outer / args.outer)))
    threads = args.cpu_threads or usable_cpus()
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(1, inner, outer, 0, outer, INNER_FACTOR, OUTER_FACTOR, ik, iv, ok, ov)
    hik, hiv, hok, hov = ik.download(), iv.download(), ok.download(), ov.download()
    sums = hj.column_sums(ok, outer, OUTER_FACTOR, INNER_FACTOR)
    for c in (ik, iv, ok, ov):
        c.free()
    # AVX-512 histogram / partition / probe where the host has it.  The partition exists in two forms - tuples moved
    # by scalar stores into the write-combining lines, or by the reference's conflict-serialised vector scatter
    # (phj.cpp:1099-1160) -: the faster one on THIS host (timed on a 64 M-tuple sample) is the baseline
    simd, calib = 0, ""
    n_s = min(outer, 64_000_000)
    m_s = max(1, int(inner * (n_s / outer)))
    rates = {}
    for mode in (1, 2):
        if O.set_simd(mode) != mode:
            break
        t = O.Timing()
        O.phj(hik[:m_s], hiv[:m_s], hok[:n_s], hov[:n_s], threads=threads, timing=t)
        rates[mode] = n_s / t.seconds / 1e9
    if rates:
        simd = max(rates, key=rates.get)
        calib = "; partition by %s (%.2f Gtuples/s on a 64 M-tuple sample, the other form %.2f)" % (
            "conflict-serialised vector scatter (phj.cpp:1099-1160)" if simd == 2 else
            "scalar stores into the write-combining lines", rates[simd], rates[3 - simd] if 3 - simd in rates else 0.0)
    O.set_simd(simd)

    def one(which):
        tm = O.Timing()
        if which == "npj":
            res = O.npj(hik, hiv, hok, hov, threads=threads, load=0.90, timing=tm)     # npj.cpp:944
        elif which == "cpra":
            res = O.cpra(hik, hiv, hok, hov, threads=threads, timing=tm)
        else:
            res = O.phj(hik, hiv, hok, hov, threads=threads, timing=tm)
        ok_ = res == (outer, sums[0], sums[1], sums[2])
        return {"value": outer / tm.seconds / 1e9, "unit": "Gtuples/s", "cores": threads,
                "kind": "port",
                "sample": "%s |R|=%d join |S|=%d (same generator, 1/%g of the per-GPU workload), "
                          "oracle/hj_oracle.c pthreads restatement of %s with %s operators%s, %d threads "
                          "(%d CPUs online, cgroup quota applied), %.3f s, checksum %s"
                          % (which, inner, outer, args.outer / outer, "run() (npj.cpp:769-927, load 0.90)" if which == "npj" else "run_hj",
                             "AVX-512 (oracle/hj_oracle_avx512.c)" if simd else "scalar", calib if which != "npj" else "", threads, os.cpu_count() or 0,
                             tm.seconds, "ok" if ok_ else "MISMATCH"),
                "seconds": tm.seconds}

    out = one(algo)
    others = {}
    for which in more:
        try:
            others[which] = one(which)
        except Exception as ex:          # reported, never required
            others[which] = {"value": None, "unit": "Gtuples/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
    O.set_simd(False)
    if others:
        out["others"] = others
    return out


def cpu_baseline_config0(hj, args):
    """BASELINE.json configs[0]: `./npj 64 1000000 16000000` (npj.cpp:929-935: 64 threads, 1 M probe tuples, 16 M
    build tuples = 16 copies per key) as the oracle's restatement of run() (npj.cpp:769-927, load 0.90,
    npj.cpp:944) on this host; threads = min(64, usable CPUs) (npj.cpp:952 asserts threads <= hardware threads).
    The same relations are joined on the GPU and the aggregates compared."""
    from oracle import oracle as O
    threads = min(64, args.cpu_threads or usable_cpus())
    ik, iv, ok, ov = O.generate(1_000_000, 16_000_000, seed=1)
    tm = O.Timing()
    res = O.npj(ik, iv, ok, ov, threads=threads, load=0.90, timing=tm)
    rk, rv, sk, sv = (hj.column(c) for c in (ik, iv, ok, ov))
    gpu = hj.npj(rk, rv, len(ik), sk, sv, len(ok))
    gpu_ms = hj.stats()["ms_total"]
    for c in (rk, rv, sk, sv):
        c.free()
    return {"value": len(ok) / tm.seconds / 1e9, "unit": "Gtuples/s (probe side)", "cores": threads, "kind": "port",
            "seconds": tm.seconds, "join_tuples": res[0],
            "phase_seconds": {"init_build": tm.seconds_phase[0], "probe": tm.seconds_phase[1], "close_gaps": tm.seconds_phase[2]},
            "sample": "./npj %d 1000000 16000000: oracle/hj_oracle.c restatement of run() (npj.cpp:769-927), load 0.90, "
                      "relations of the T = 1 generator (cpra2.cpp:1578-1696, seed 1), J = %d, %.4f s; the same relations "
                      "on the GPU (hjgpu_npj, %.3f ms): aggregates %s" % (threads, res[0], tm.seconds, gpu_ms,
                                                                         "equal" if gpu == res else "DIFFER"),
            "gpu_ms": gpu_ms, "checksum_ok": gpu == res}


def roof(bytes_per_step, ms, launches, stream_read_gbs, bound="hbm"):
    """roofline entry of one kernel: algorithmic bytes / measured time"""
    if ms <= 0 or launches <= 0:
        return None
    gbs = bytes_per_step / (ms * 1e-3) / 1e9
    return {"bound": bound, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
            "frac_of_stream_read": round(gbs / stream_read_gbs, 4) if stream_read_gbs else None,
            "avg_launch_ms": round(ms / launches, 4), "launches_per_step": launches,
            "algorithmic_bytes_per_launch": int(bytes_per_step / launches)}


def attach_traffic(H, kernels, args, n_gpus):
    """HBM bytes per launch from the PMC passes of tools/collect_traffic.py: only from TRAFFIC_FILE, only when it
    was measured with the kernels that are running now and on this workload."""
    if (args.inner, args.outer, args.algo, n_gpus, args.zipf) != (64_000_000, 1_000_000_000, "phj", 1, 0.0):
        return "none: PMC traffic exists for the default workload only"
    if not os.path.exists(TRAFFIC_FILE):
        return "none: %s is missing" % os.path.relpath(TRAFFIC_FILE, ROOT)
    t = json.load(open(TRAFFIC_FILE))
    name = os.path.relpath(TRAFFIC_FILE, ROOT)
    if t.get("kernel_hash") != H.kernel_hash():
        return "refused: %s was measured with kernels %s, this library is %s" % (name, t.get("kernel_hash"), H.kernel_hash())
    if t.get("bench_args"):
        return "refused: %s was measured with bench.py %s" % (name, " ".join(t["bench_args"]))
    for kname, entry in kernels.items():
        hit = [v for k, v in t["kernels"].items() if k.startswith(kname)]
        if hit:
            # several template instances (pass 1 / pass 2) share a launch name prefix
            entry["traffic"] = int(sum(h["hbm_bytes_per_launch"] * h["launches_seen"] for h in hit)
                                   / max(1, sum(h["launches_seen"] for h in hit)))
    return "%s (kernels %s)" % (name, t["kernel_hash"])


def workspace_report(hj):
    """what the context spent growing its workspace before the timed region, and its placement search for the probe side's
    pass-1 twin (hjgpu_stats.ms_reserve / placement_*): candidates allocated and filled, the kept block's fill rate, whether
    the search's budget (option placement_ms, 500 ms) ended it"""
    st = hj.stats()
    fill = st["placement_fill_ms"]
    return {"ms_reserve": round(st["ms_reserve"], 1), "candidates_tried": int(st["placement_tried"]),
            "chosen_fill_ms": round(fill, 3), "chosen_fill_TBs": round(st["placement_bytes"] / (fill * 1e-3) / 1e12, 2) if fill > 0 else None,
            "chosen_is_fast_kind": bool(fill > 0 and st["placement_bytes"] / (fill * 1e-3) >= 5.5e12),
            "search_ms": round(st["placement_search_ms"], 1),
            "search_timeboxed": bool(st["placement_timeboxed"]), "search_budget_ms": 500}


def attach_secondary_traffic(H, leg, entries):
    """entries: {roofline object: (kernel name prefix, "mean" | "sum" over the template instances)}; returns the source text"""
    fname, want_args = SECONDARY_TRAFFIC[leg]
    path = os.path.join(ROOT, "profiles", fname)
    for entry, _ in entries:
        if entry is not None:
            entry["traffic"] = None
    if not os.path.exists(path):
        return "none: profiles/%s is missing" % fname
    t = json.load(open(path))
    if t.get("kernel_hash") != H.kernel_hash():
        return "refused: profiles/%s was measured with kernels %s, this library is %s" % (fname, t.get("kernel_hash"), H.kernel_hash())
    if list(t.get("bench_args") or []) != want_args:
        return "refused: profiles/%s was measured with %r, this leg is %r" % (fname, t.get("bench_args"), want_args)
    for entry, (prefix, how) in entries:
        hit = [v for k, v in t["kernels"].items() if k.startswith(prefix)]
        if entry is None or not hit:
            continue
        if how == "sum":          # every instance runs once per join (the two launches of a _UNIQUE join)
            entry["traffic"] = int(sum(h["hbm_bytes_per_launch"] for h in hit))
        else:
            entry["traffic"] = int(sum(h["hbm_bytes_per_launch"] * h["launches_seen"] for h in hit) / max(1, sum(h["launches_seen"] for h in hit)))
    return "profiles/%s (kernels %s)" % (fname, t["kernel_hash"])


def cpra_multi_leg(args, H, torch, dist, comm, hj, dev, rank, n_gpus):
    """BASELINE configs[4] inside the N > 1 line: CPRA with BOTH relations chunked over the ranks (cpra2.cpp:1757-1827 the
    own-chunk partitioning, 1868-1872 ownership, 1891-1959 the gather = one all-to-all-v over xGMI), 128 M build and
    2 G probe tuples per rank (|R| = 1 G x |S| = 16 G at 8 GPUs), default slices.  Every rank calls this (collectives).
    --rehearse-solo: every process joins a self-contained pair of relations through its own one-rank communicator."""
    inner, outer = args.configs4_inner, args.configs4_outer
    solo = args.rehearse_solo
    stream = torch.cuda.current_stream().cuda_stream
    cols = [torch.empty(n + 4, dtype=torch.int32, device=dev) for n in (inner, inner, outer, outer)]
    rk, rv, sk, sv = (c.data_ptr() for c in cols)
    if solo:
        hj.generate_range(1 + rank, inner, outer, 0, inner, 0, outer, INNER_FACTOR, OUTER_FACTOR, rk, rv, sk, sv, stream)
    else:
        hj.generate_range(1, inner * n_gpus, outer * n_gpus, rank * inner, inner, rank * outer, outer,
                          INNER_FACTOR, OUTER_FACTOR, rk, rv, sk, sv, stream)
    sums = hj.column_sums(sk, outer, OUTER_FACTOR, INNER_FACTOR, stream)
    expect = sum_over_ranks(dist, torch, [outer, sums[0], sums[1], sums[2]])
    shards = [(rk, rv, inner, sk, sv, outer)]
    prm = H.PhjParams()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one():
        res, ms = comm.cpra_multi(shards, prm, args.exchange_slices)
        res = list(res)
        if solo:
            res = sum_over_ranks(dist, torch, res)
        return res, ms

    try:
        ok = one()[0] == expect                        # warm-up: buffers, workspaces, RCCL's channels
        sync()
        t0 = time.perf_counter()
        steps = []
        for _ in range(args.configs4_steps):
            res, ms = one()
            ok = ok and res == expect
            steps.append(ms)
        sync()
    except H.HjGpuError as ex:
        die_of_comm_error(rank, ex, "secondary.cpra_multi")
    elapsed = max_over_ranks(dist, torch, time.perf_counter() - t0)
    k = len(steps)
    mine = {"ms_wall_inside_library": sum(m["ms_wall"] for m in steps) / k,
            "ms_exchange": sum(m["ms_exchange"] for m in steps) / k,
            "ms_partition": sum(m["ms_partition"] for m in steps) / k,
            "ms_joins_waited_for_exchange": sum(m["ms_exchange_wait"] for m in steps) / k,
            "MB_sent": sum(m["bytes_sent"] for m in steps) / k / 1e6,
            "self_copies": sum(m["self_copies"] for m in steps) / k,
            "ms_join_kernel_last_batch": steps[-1]["join"]["ms_join"],
            "fanout": [steps[-1]["join"]["fanout1"], steps[-1]["join"]["fanout2"]]}
    mine["exchange_GBs_sent"] = mine["MB_sent"] / mine["ms_exchange"] if mine["ms_exchange"] > 0 else 0.0
    every = gather_objects(dist, mine)
    ms_step = elapsed / k * 1e3
    total_outer = outer * n_gpus
    keys = ["ms_wall_inside_library", "ms_exchange", "ms_partition", "ms_joins_waited_for_exchange", "MB_sent", "exchange_GBs_sent",
            "ms_join_kernel_last_batch"]
    leg = {"workload": "CPRA |R|=%d x |S|=%d over %d GPU(s): %d build and %d probe tuples per rank (BASELINE configs[4]'s shape), both "
                       "sides chunked, exchange-level fan-out = pass 1, one all-to-all-v per slice, %s slices"
                       % (inner * n_gpus, total_outer, n_gpus, inner, outer, args.exchange_slices or "default"),
           "steps": k, "ms_per_step": round(ms_step, 4), "gtuples_per_s": round(total_outer / ms_step / 1e6, 3),
           "gtuples_per_s_per_gpu": round(outer / ms_step / 1e6, 3), "checksum_ok": bool(ok),
           "fanout_of_rank0": mine["fanout"],
           "rank0": {k2: round(mine[k2], 4) for k2 in keys},
           "all_ranks": {k2: spread([e[k2] for e in every]) for k2 in keys},
           # HBM bytes per tuple and rank: sender 4 + 16, receiver 8 + 16 + 8, exchange 16 (G - 1) / G
           "hbm_bytes_per_tuple": 52 + 16.0 * (n_gpus - 1) / n_gpus if not solo else 52}
    if solo:
        leg["rehearsal"] = "--rehearse-solo: one-rank communicators, self-contained relations per process"
    del cols
    torch.cuda.empty_cache()
    return leg


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        relaunch_under_torchrun(args)
    # stdout carries ONE line, the result: whatever libraries print there on the way (RCCL's version banner at
    # communicator creation goes to stdout) is sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = max(args.gpus, world)

    import torch
    import hash_join_codes_knl_amd as H
    if args.rehearse_solo:
        local_rank = 0                     # every process on the one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    comm = None
    multi = n_gpus > 1 or args.force_dist
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=max(world, 1))      # control plane only
        try:
            if args.rehearse_solo:
                os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
                comm = H.HjComm.rank(local_rank, 1, 0, H.HjComm.new_id())          # a world of one per process
            else:
                comm = connect_ranks(dist, H, local_rank, rank, max(world, 1))     # data plane: RCCL from C++
        except H.HjGpuError as ex:
            die_of_comm_error(rank, ex, "creating the communicator")
        comm.set_option("timeout_ms", args.comm_timeout_ms)
        for nv in args.comm_option:
            name, _, value = nv.partition("=")
            comm.set_option(name, int(value))
        if args.ring_broadcast:
            comm.set_option("ring_broadcast", 1)
        if args.reserve_cus >= 0:
            comm.set_option("reserve_cus", args.reserve_cus)
        hj = comm.ctx[0]
    else:
        hj = H.HjGpu(local_rank)
    # N = 1: the headline is the blocking join with the library's DEFAULT store policy (every store non-temporal: what any caller gets).
    # --solo: option "solo" for the headline (plain partial-line stores; the line says so in config.solo_stores); the default line
    # reports that form beside it (secondary.phj_solo).  The multi-GPU joins and every enqueue-only form never use it.
    solo = not multi and not args.enqueue_only and args.solo and not args.no_solo
    if solo:
        hj.set_option("solo", "1")
    for o in args.option:
        name, _, value = o.partition("=")
        hj.set_option(name, value)
    info = hj.device_info()
    inner, outer = args.inner, args.outer
    outer_total = outer * n_gpus

    # ---- relations resident in HBM (torch owns the memory; the library borrows pointers)
    def col(n):
        return torch.empty(n + 4, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # CPRA across GPUs (BASELINE configs[4]): every rank owns a chunk of BOTH relations and the
    # tuples are co-partitioned with one all-to-all; PHJ / NPJ: R replicated, S sharded.
    copart = multi and args.algo == "cpra"
    inner_total = inner * n_gpus if copart else inner
    rk, rv, sk, sv = col(inner), col(inner), col(outer), col(outer)
    if copart and args.rehearse_solo:
        sys.exit("--rehearse-solo joins replicated build sides only (every process needs the whole build side)")
    if copart:
        hj.generate_range(1, inner_total, outer_total, rank * inner, inner, rank * outer, outer,
                          INNER_FACTOR, OUTER_FACTOR, rk.data_ptr(), rv.data_ptr(), sk.data_ptr(),
                          sv.data_ptr(), stream)
    else:
        # the build side lives on rank 0 (the others only hold room for it); every rank generates its probe shard
        hj.generate_zipf(1, inner, outer_total, 0, inner, rank * outer, outer, INNER_FACTOR, OUTER_FACTOR,
                         args.zipf, rk.data_ptr(), rv.data_ptr(), sk.data_ptr(), sv.data_ptr(), stream)
    sums = hj.column_sums(sk.data_ptr(), outer, OUTER_FACTOR, INNER_FACTOR, stream)
    expect_local = [outer, sums[0], sums[1], sums[2]]
    expect_global = sum_over_ranks(dist, torch, expect_local)

    hj.reserve(inner, outer)
    prm = H.PhjParams(fanout1=args.fanout1, fanout2=args.fanout2)
    nprm = H.NpjParams()                       # library default load factor (0.25)
    d_result = torch.zeros(4, dtype=torch.int64, device=dev)
    if multi:
        # replicated build: only the root's build columns are read, every step replicates them again (measured,
        # not assumed); co-partitioned: the rank's chunk of both relations
        root_cols = (rk.data_ptr(), rv.data_ptr()) if (copart or rank == 0 or args.rehearse_solo) else (None, None)
        shards = [(root_cols[0], root_cols[1], inner, sk.data_ptr(), sv.data_ptr(), outer)]

    last = {}

    def step():
        if multi:
            if copart:
                res, ms = comm.cpra_multi(shards, prm, args.exchange_slices)
            elif args.algo == "npj":
                res, ms = comm.npj_multi(shards, 0, nprm)
            else:
                res, ms = comm.phj_multi(shards, 0, prm)
            if args.rehearse_solo:         # the solo communicators know their own shard only: add the ranks' results up
                res = sum_over_ranks(dist, torch, list(res))
            last["result"], last["multi"] = list(res), ms
            return
        s = torch.cuda.current_stream().cuda_stream
        a = (rk.data_ptr(), rv.data_ptr(), inner, sk.data_ptr(), sv.data_ptr(), outer)
        # One step = one complete join through the BLOCKING entry point - the counterpart of the reference's run() / run_hj(), which
        # return when the join is done (npj.cpp:861-918) - on this process's only stream, with option "solo" (see above): its
        # partial-line and row stores stay plain (DESIGN section 3 "Round 5").  The enqueue-only forms, which may run beside other
        # streams' work and therefore write everything non-temporal, are timed as secondary.phj_enqueue_only.
        if args.enqueue_only:
            if args.algo == "phj":
                hj.phj_async(*a, prm, d_result.data_ptr(), s)
            elif args.algo == "cpra":
                hj.cpra_async(*a, prm, d_result.data_ptr(), s)
            else:
                hj.npj_async(*a, nprm, d_result.data_ptr(), s)
            return
        if args.algo == "phj":
            last["result"] = list(hj.phj(*a, prm, stream=s))
        elif args.algo == "cpra":
            last["result"] = list(hj.cpra(*a, prm, stream=s))
        else:
            last["result"] = list(hj.npj(*a, nprm, stream=s))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    phases = ["ms_total", "ms_histogram", "ms_plan", "ms_scatter1", "ms_scatter2", "ms_join",
              "ms_build", "ms_close_gaps", "ms_inner_wait", "ms_scatter0"]
    per_step = {p: [] for p in phases}
    multi_steps = []
    preflight = None
    if multi:
        # before anything is timed: every collective the joins use, verified word for word, and the point-to-point
        # rate to every peer ("measure link bandwidth first"); a wrong byte or a rank that never
        # arrives ends the run here, with a message, instead of hanging the timed region
        try:
            if args.preflight_bytes > 0:
                preflight = comm.preflight(args.preflight_bytes)
            for _ in range(args.warmup):
                step()
        except H.HjGpuError as ex:
            die_of_comm_error(rank, ex, "preflight / warm-up")
    else:
        for _ in range(args.warmup):
            step()
    barrier()
    # empirical streaming-read ceiling of this box: a plain 16-byte-load sweep of the
    # 4 GB probe-key column (nothing computed), outside the timed region
    best = 1e9
    for _ in range(4):
        best = min(best, hj.stream_read_ms(sk.data_ptr(), 4 * outer // 65536 * 65536, stream))
    stream_read_gbs = (4 * outer // 65536 * 65536) / (best * 1e-3) / 1e9 if outer >= 65536 else None
    barrier()
    results_ok = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        try:
            step()
        except H.HjGpuError as ex:
            if multi:
                die_of_comm_error(rank, ex, "timed step")
            raise
        if multi:
            ms = last["multi"]
            multi_steps.append(ms)
            for p in phases:
                per_step[p].append(ms["join"][p])
            results_ok = results_ok and last["result"] == expect_global
        else:
            st = hj.stats()             # hipEvent spans of this step's kernels (same stream)
            for p in phases:
                per_step[p].append(st[p])
            if not args.enqueue_only:   # the blocking call returned this step's aggregates: every step is checked
                results_ok = results_ok and last["result"] == expect_global
    barrier()
    elapsed = max_over_ranks(dist, torch, time.perf_counter() - t0)
    got = last["result"] if (multi or not args.enqueue_only) else [int(x) & MASK64 for x in d_result.tolist()]
    # the analytic aggregates assume unique build keys, i.e. at least as many probe as build tuples
    checksum_ok = (got == expect_global and results_ok) if outer_total >= inner_total else None

    ms_per_step = elapsed / args.steps * 1e3
    value = outer_total / (elapsed / args.steps) / 1e9
    avg = {p: sum(v) / max(1, len(v)) for p, v in per_step.items()}
    st = multi_steps[-1]["join"] if multi else hj.stats()
    # tuples the kernels of one step read on this rank, and the local join calls they were read by
    if multi:
        n_tuples = sum(m["tuples_joined"] for m in multi_steps) / args.steps
        jps = max(1, round(sum(m["joins"] for m in multi_steps) / args.steps))
    else:
        n_tuples, jps = inner + outer, 1

    kernels = {}
    if args.algo in ("phj", "cpra"):
        two = st["fanout2"] > 1
        groups = int(st.get("groups", 0))       # a grouped plan: pass 0 over both relations, then `groups` two-pass joins
        # a prepared build side (co-partitioned CPRA) is one call for R and one per probe slice: one launch each
        per_call = 1 if copart else 2
        kernels["hist2_kernel"] = roof(4 * n_tuples, avg["ms_histogram"], 2 * groups if groups else per_call * jps, stream_read_gbs)
        if groups:       # (ms_scatter0 is the whole pass-0 operator, its histogram included: the fraction errs low)
            kernels["scatter_kernel"] = roof(3 * 16 * n_tuples, avg["ms_scatter0"] + avg["ms_scatter1"] + avg["ms_scatter2"],
                                             2 + 4 * groups, stream_read_gbs)
        else:
            kernels["scatter_kernel"] = roof((2 if two else 1) * 16 * n_tuples, avg["ms_scatter1"] + avg["ms_scatter2"],
                                             (2 if two else 1) * per_call * jps, stream_read_gbs)
        kernels["join_kernel"] = roof(8 * n_tuples, avg["ms_join"], max(1, jps - 1) if copart else jps, stream_read_gbs)
    else:
        kernels["npj_build_kernel"] = roof(8 * inner + 8 * st["buckets"] + 8 * inner, avg["ms_build"], 1, stream_read_gbs)
        kernels["npj_probe_kernel"] = roof(16 * outer, avg["ms_join"], 1, stream_read_gbs)
    join_ms = avg["ms_join"]
    kernels = {k: v for k, v in kernels.items() if v}
    traffic_src = attach_traffic(H, kernels, args, n_gpus)
    dominant = max(kernels, key=lambda k: kernels[k]["avg_launch_ms"] * kernels[k]["launches_per_step"])
    roofline = dict(kernels[dominant])
    roofline["kernel"] = dominant

    transport = "RCCL from C++ (hjgpu_%s_multi)" % ("cpra" if copart else args.algo)
    if copart:
        parallelism = ("both sides chunked over %d GPU(s), %s: own-chunk partitioning, counts all-gather, all-to-all-v "
                       "(probe side in %d slices, transfers overlapped with partitioning and local PHJ), all-reduce"
                       % (n_gpus, transport, args.exchange_slices or (1 if max(world, 1) == 1 else 4)))
    elif multi:
        parallelism = "probe side sharded over %d GPU(s), build side replicated each step by %s, %s, overlapped with " \
                      "probe-side partitioning, all-reduce" % (n_gpus, "one ncclBroadcast" if args.ring_broadcast else
                                                               "scatter + all-gather", transport)
    else:
        parallelism = "probe side sharded over 1 GPU(s), build side local"
    out = {
        "metric": "probe Gtuples/s + % HBM roofline, PHJ |R|=64M join |S|=1G, 1/2/4/8 GPU",
        "value": round(value, 3), "unit": "Gtuples/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        # the host's share of a step: wall clock per step minus the device time between the join's first and last event (N = 1: the
        # blocking call's enqueue lead, its copy of the state block and the synchronisation; `value` is priced on the wall clock)
        "ms_host_per_step": round(ms_per_step - avg["ms_total"], 4) if not multi else None,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s end-to-end (histogram + %s + LDS build/probe, aggregate output), "
                               "%s unique 32-bit keys, |R|=%d %s, |S|=%d per GPU, selectivity 1"
                               % (args.algo.upper(), ("3 scatter passes (%d groups)" % st["groups"]) if st.get("groups") else
                                  "2 scatter passes" if st["fanout2"] > 1 else "1 scatter pass",
                                  "uniform" if args.zipf <= 0 else "Zipf(%g) probe side," % args.zipf, inner,
                                  "per GPU (co-partitioned)" if copart else "replicated", outer),
                   "algorithm": args.algo, "inner_tuples": inner, "outer_tuples_per_gpu": outer,
                   "outer_tuples_total": outer_total,
                   "fanout": [st["fanout1"], st["fanout2"]], "groups": int(st.get("groups", 0)),
                   # option "solo" (blocking joins of a process that runs nothing else on the device): partial-line stores plain
                   "solo_stores": bool(solo),
                   "parallelism": parallelism},
        "roofline": roofline,
        "roofline_kernels": kernels,
        "traffic_source": traffic_src,
        "kernel_hash": H.kernel_hash(), "library_hash": H.library_hash(),
        "join_phase": {"gtuples_per_s_per_gpu": round(outer / (join_ms * 1e-3) / 1e9, 2) if join_ms > 0 else None,
                       "ms": round(join_ms, 4),
                       "hbm_read_frac": round(8 * n_tuples / (join_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if join_ms > 0 else None},
        "phase_ms": {k: round(v, 4) for k, v in avg.items()},
        "phase_ms_min": {k: round(min(v), 4) for k, v in per_step.items() if v},
        "phase_ms_max": {k: round(max(v), 4) for k, v in per_step.items() if v},
        "empirical_stream_read_GBs": round(stream_read_gbs, 1) if stream_read_gbs else None,
        "checksum_ok": checksum_ok,
        "result": {"count": got[0], "sum_keys": got[1], "sum_outer_vals": got[2], "sum_inner_vals": got[3]},
        "device": info["name"], "arch": info["arch"],
        # wall clock this context spent growing its workspace before the timed region (allocations + the placement search
        # for the probe side's pass-1 twin: up to 12 candidate blocks of 8.5 GB held and filled twice, hjgpu_stats.ms_reserve)
        "workspace": workspace_report(hj),
    }
    if args.rehearse_solo:
        out["rehearsal"] = "--rehearse-solo: %d processes on ONE GPU, each with a one-rank RCCL communicator; the numbers mean nothing" % world
    if multi:
        k = len(multi_steps)
        mine = {  # this rank's view, averaged over the timed steps
            "rank": rank,
            "ms": sum(m["ms_exchange"] for m in multi_steps) / k,
            "ms_partition": sum(m["ms_partition"] for m in multi_steps) / k,
            "ms_joins_waited_for_it": sum(m["ms_exchange_wait"] for m in multi_steps) / k,
            "MB_sent": sum(m["bytes_sent"] for m in multi_steps) / k / 1e6,
            "ms_wall_inside_library": sum(m["ms_wall"] for m in multi_steps) / k,
            "ms_join_kernel": avg["ms_join"], "ms_scatter": avg["ms_scatter1"] + avg["ms_scatter2"],
            "ms_histogram": avg["ms_histogram"],
            "rccl": comm.info(), "preflight": preflight}
        every = gather_objects(dist, mine)
        keys = ["ms", "ms_partition", "ms_joins_waited_for_it", "MB_sent", "ms_wall_inside_library", "ms_join_kernel",
                "ms_scatter", "ms_histogram"]
        out["exchange"] = {k2: round(mine[k2], 4) for k2 in keys}           # rank 0's view ...
        out["exchange"]["local_join_calls_measured"] = jps
        out["exchange"]["all_ranks"] = {k2: spread([e[k2] for e in every]) for k2 in keys}     # ... and min / max / mean over ALL ranks
        # what RCCL itself says the world is (ncclGetVersion, ncclCommCount, ncclCommUserRank of every rank's
        # communicator): proof that N ranks talked over RCCL and not N replicas next to each other
        infos = [e["rccl"] for e in every]
        out["rccl"] = {"transport": infos[0]["transport"], "version": infos[0]["rccl_version"],
                       "nranks_by_ncclCommCount": sorted({i["rccl_nranks"] for i in infos}),
                       "ranks_by_ncclCommUserRank": sorted(i["rccl_rank"] for i in infos),
                       "devices_by_ncclCommCuDevice": [i["rccl_device"] for i in infos],
                       "timeout_ms": infos[0]["timeout_ms"], "aborted": sorted({i["aborted"] for i in infos})}
        if preflight:
            out["exchange"]["preflight"] = {
                "collectives_verified": all(e["preflight"] and e["preflight"]["ok_all_gather"] and e["preflight"]["ok_all_to_all"]
                                            and e["preflight"]["ok_all_reduce"] for e in every),
                "bytes_per_peer": preflight["link_bytes"],
                # link_GBs[src][dst]: GB/s of `bytes_per_peer` from src to dst while every rank sends to its peer at the
                # same distance (all pairs' own xGMI links at once)
                "link_GBs": [e["preflight"]["link_GBs"] for e in every],
                "all_to_all_GBs_sent_per_rank": [round(e["preflight"]["all_to_all_GBs"], 2) for e in every],
                "ms_first_collectives_1MB": {"all_gather": round(preflight["ms_all_gather"], 3),
                                             "all_to_all_v": round(preflight["ms_all_to_all"], 3),
                                             "all_reduce": round(preflight["ms_all_reduce"], 3)}}

    # ---- BASELINE configs[4] beside configs[3], N > 1 (one driver command measures both multi-GPU configurations) -----
    if (multi and args.algo == "phj" and not args.no_secondary and args.zipf <= 0
            and (args.inner, args.outer) == (64_000_000, 1_000_000_000)):
        del shards
        rk = rv = sk = sv = None                       # the headline's columns make room (torch frees, the library keeps its workspaces)
        torch.cuda.empty_cache()
        leg = cpra_multi_leg(args, H, torch, dist, comm, hj, dev, rank, n_gpus)
        out.setdefault("secondary", {})["cpra_multi"] = leg

    # ---- the other BASELINE configurations, same process, N = 1 ---------------------------------------------------
    extras = (not multi and rank == 0 and not args.no_secondary and args.algo == "phj" and args.zipf <= 0
              and (args.inner, args.outer) == (64_000_000, 1_000_000_000))
    if extras:
        def time_steps(fn, warm=1, steps=5, ctx=None):
            """ms per call (host clock over `steps` back-to-back enqueues + one synchronise) and the per-phase stats"""
            ctx = ctx or hj
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            acc = {}
            for _ in range(steps):
                fn()
                for key, val in ctx.stats().items():          # waits for this call's last event
                    acc[key] = acc.get(key, 0.0) + val
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / steps * 1e3, {key: val / steps for key, val in acc.items()}

        a = (rk.data_ptr(), rv.data_ptr(), inner, sk.data_ptr(), sv.data_ptr(), outer)
        sec = {}
        # configs[1]: NPJ build + probe on one GPU
        ms, ph = time_steps(lambda: hj.npj_async(*a, nprm, d_result.data_ptr(), stream))
        ok_ = [int(x) & MASK64 for x in d_result.tolist()] == expect_local
        # what the memory system gives for NPJ's access shape: 10^9 independent pseudo-random 64-byte line reads out of a
        # 2 GiB buffer (the table's size at load 0.25), four lanes per line as in the probe, nothing else computed
        table_bytes = 2 << 30
        scratch_t = torch.empty(table_bytes // 4, dtype=torch.int32, device=dev)
        rl_ms = min(hj.random_line_read_ms(scratch_t.data_ptr(), table_bytes, outer, stream) for _ in range(3))
        # ... and for the build's: `inner` independent random 8-byte CAS into the same (zeroed) 2 GiB, four in flight per lane,
        # without and with a look at the bucket first (the build looks: its table is line-hashed, every key of a line starts
        # at the line's first bucket).  The clear of the table is part of the build phase (npj.cpp:865-868): timed apart.
        cas_ms = min(hj.random_cas_ms(scratch_t.data_ptr(), table_bytes, inner, 4, False, stream) for _ in range(3))
        cas_look_ms = min(hj.random_cas_ms(scratch_t.data_ptr(), table_bytes, inner, 4, True, stream) for _ in range(3))
        clr = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            scratch_t.zero_()
            e1.record()
            e1.synchronize()
            clr.append(e0.elapsed_time(e1))
        clear_ms = min(clr)
        del scratch_t
        sec["npj"] = {"workload": "NPJ |R|=%d join |S|=%d, global line-hashed table, load %.2f (%d buckets)"
                                  % (inner, outer, 0.25, int(ph["buckets"])),
                      "ms_per_step": round(ms, 4), "gtuples_per_s": round(outer / ms / 1e6, 2),
                      "ms_build": round(ph["ms_build"], 4), "ms_probe": round(ph["ms_join"], 4), "checksum_ok": ok_,
                      # 16 B per probe tuple (8 B streamed + one 8-byte bucket); a bucket costs its whole
                      # 64-byte line at the memory side: 8 + 64 B "sector-granular"
                      "roofline_probe": roof(16 * outer, ph["ms_join"], 1, stream_read_gbs),
                      "roofline_probe_line_granular": roof(72 * outer, ph["ms_join"], 1, stream_read_gbs),
                      # the empirical ceiling of that shape on this box: random 64-byte line reads per second
                      "random_line_read_ceiling": {"ms_per_1e9_lines": round(rl_ms * 1e9 / outer, 3),
                                                   "Glines_per_s": round(outer / rl_ms / 1e6, 2),
                                                   "probe_frac_of_it": round(rl_ms / ph["ms_join"], 4)},
                      "roofline_build": roof(8 * inner + 8 * ph["buckets"] + 8 * inner, ph["ms_build"], 1, stream_read_gbs),
                      # the build's own ceiling: the memory system serves ~17 G independent returning 8-byte atomics per
                      # second whatever the number in flight per lane (tools/npj_build_ceiling.py); build phase = table
                      # clear + one claim per build tuple
                      "random_cas_ceiling": {"ms_per_build_side": round(cas_ms, 3), "Gcas_per_s": round(inner / cas_ms / 1e6, 2),
                                             "ms_with_a_look_first": round(cas_look_ms, 3), "ms_table_clear": round(clear_ms, 3),
                                             "build_claims_ms": round(ph["ms_build"] - clear_ms, 3),
                                             "build_claims_frac_of_it": round(cas_ms / max(ph["ms_build"] - clear_ms, 1e-3), 4)}}
        # one-GPU CPRA: 8 chunks partitioned independently (cpra2.cpp:1757-1827), gathered in place
        cprm = H.PhjParams(chunks=8)
        ms, ph = time_steps(lambda: hj.cpra_async(*a, cprm, d_result.data_ptr(), stream))
        ok_ = [int(x) & MASK64 for x in d_result.tolist()] == expect_local
        sec["cpra"] = {"workload": "CPRA |R|=%d join |S|=%d, 8 chunks, fan-out %d x %d" % (inner, outer, int(ph["fanout1"]), int(ph["fanout2"])),
                       "ms_per_step": round(ms, 4), "gtuples_per_s": round(outer / ms / 1e6, 2), "checksum_ok": ok_,
                       "phase_ms": {k2: round(ph[k2], 4) for k2 in phases if k2 in ph},
                       "roofline_scatter": roof(2 * 16 * (inner + outer), ph["ms_scatter1"] + ph["ms_scatter2"], 4, stream_read_gbs),
                       "roofline_join": roof(8 * (inner + outer), ph["ms_join"], 1, stream_read_gbs)}
        # the reference's -D_UNIQUE build (npj.cpp:288-290, phj.cpp:635-637): a probe tuple reports its first
        # match.  Same relations (unique build keys: same result), HJGPU_FLAG_UNIQUE
        uprm = H.PhjParams(fanout1=args.fanout1, fanout2=args.fanout2, flags=H.FLAG_UNIQUE)
        ms, ph = time_steps(lambda: hj.phj_async(*a, uprm, d_result.data_ptr(), stream))
        ok_ = [int(x) & MASK64 for x in d_result.tolist()] == expect_local
        sec["phj_unique"] = {"workload": "PHJ |R|=%d join |S|=%d with HJGPU_FLAG_UNIQUE (-D_UNIQUE: first match only)" % (inner, outer),
                             "ms_per_step": round(ms, 4), "gtuples_per_s": round(outer / ms / 1e6, 2), "checksum_ok": ok_,
                             "ms_join": round(ph["ms_join"], 4),
                             "join_vs_default_instance": round(ph["ms_join"] / join_ms, 4) if join_ms > 0 else None,
                             "roofline_join": roof(8 * (inner + outer), ph["ms_join"], 1, stream_read_gbs)}
        # the headline through the enqueue-only entry point: a join that may run beside other streams' work writes every K6 store
        # non-temporal (partial lines included: +0.15-0.4 ms per pass, profiles/r05_ab_stores.txt), the price of never losing one
        ms, ph = time_steps(lambda: hj.phj_async(*a, prm, d_result.data_ptr(), stream), warm=2, steps=min(args.steps, 10))
        ok_ = [int(x) & MASK64 for x in d_result.tolist()] == expect_local
        sec["phj_enqueue_only"] = {"workload": "the headline through hjgpu_phj_async (enqueue-only: partial-line stores non-temporal too)",
                                   "ms_per_step": round(ms, 4), "gtuples_per_s": round(outer / ms / 1e6, 2), "checksum_ok": ok_,
                                   "phase_ms": {k2: round(ph[k2], 4) for k2 in phases if k2 in ph}}
        # the headline with option "solo" (the caller's promise that nothing else runs on the device beside its blocking joins: K6's partial-line
        # stores plain): same context, same workspace, the option set for these steps only
        if not solo:
            hj.set_option("solo", "1")
            ms, ph = time_steps(lambda: hj.phj(*a, prm), warm=2, steps=min(args.steps, 10))
            sec["phj_solo"] = {"workload": "the headline with option solo (blocking hjgpu_phj, partial-line stores plain: a process that runs nothing else on the device)",
                               "ms_per_step": round(ms, 4), "gtuples_per_s": round(outer / ms / 1e6, 2), "checksum_ok": list(hj.phj(*a, prm)) == expect_local,
                               "phase_ms": {k2: round(ph[k2], 4) for k2 in phases if k2 in ph}}
            hj.set_option("solo", "0")
        # the headline WITHOUT the placement search (option placement=1: the first allocation is taken): what a step costs when
        # the probe side's pass-1 twin is whatever block hipMalloc returns (a context of its own, closed afterwards)
        hj1 = H.HjGpu(local_rank)
        try:
            hj1.set_option("placement", "1")
            if solo:
                hj1.set_option("solo", "1")
            hj1.reserve(inner, outer)
            ms, ph = time_steps(lambda: hj1.phj_async(*a, prm, d_result.data_ptr(), stream), warm=2, steps=min(args.steps, 10), ctx=hj1)
            ok_ = [int(x) & MASK64 for x in d_result.tolist()] == expect_local
            ws1 = hj1.stats()
            sec["phj_unplaced"] = {"workload": "the headline with option placement=1 (no search: the first allocation holds the probe side's pass-1 twin)",
                                   "ms_per_step": round(ms, 4), "gtuples_per_s": round(outer / ms / 1e6, 2), "checksum_ok": ok_,
                                   "ms_scatter1": round(ph["ms_scatter1"], 4), "ms_scatter1_headline": round(avg["ms_scatter1"], 4),
                                   "ms_reserve": round(ws1["ms_reserve"], 1)}
        finally:
            hj1.close()
        sec["npj"]["traffic_source"] = attach_secondary_traffic(H, "npj", [(sec["npj"]["roofline_probe"], ("npj_probe", "mean")),
                                                                             (sec["npj"]["roofline_probe_line_granular"], ("npj_probe", "mean")),
                                                                             (sec["npj"]["roofline_build"], ("npj_build", "mean"))])
        sec["cpra"]["traffic_source"] = attach_secondary_traffic(H, "cpra", [(sec["cpra"]["roofline_scatter"], ("scatter_kernel", "mean")),
                                                                               (sec["cpra"]["roofline_join"], ("join_kernel", "mean"))])
        sec["phj_unique"]["traffic_source"] = attach_secondary_traffic(H, "phj_unique", [(sec["phj_unique"]["roofline_join"], ("join_kernel", "sum"))])
        out["secondary"] = sec
        # rows (key, outer payload, inner payload) written through the block
        # protocol, compacted by close_gaps; priced against read + written bytes.
        # block size of the output protocol (npj.cpp:244-246 claims 65 536-row blocks per thread): a free parameter of
        # hjgpu_output.  Every wave leaves one partly filled block for close_gaps to compact, so smaller blocks mean
        # less to move (profiles/r03_ab_emit.txt: 65536 -> 0.60 ms of close_gaps, 16384 -> 0.21, 4096 -> 0.12; at 1024
        # the claims themselves cost 8 ms)
        block = 4096
        cap = ((outer + block - 1) // block + 8192 + 8) * block
        # the three result columns are written at 4096 places at once: like the library's pass-1 twin they take 3.7 to
        # 4.3 ms to fill depending on WHICH allocation they are (tools/rows_luck.py, profiles/r03_rows_luck.txt), so they
        # come from the library's placement-aware allocator (hjgpu_malloc_placed), outside the timed calls
        jcols = [hj.column(cap, placed=True) for _ in range(3)]
        jk, jo, ji = (torch.empty(0, dtype=torch.int32, device=dev) for _ in range(3))
        # MEAN over as many steps as the headline (after one untimed call), min / max beside it - not a best-of.
        # Two policies: "materialized_default" = the library's default (result rows through non-temporal stores: what any caller gets),
        # "materialized" = option solo (plain rows: a process that runs nothing else on the device; the form of rounds 2-5's line)
        rw = 8 * n_tuples
        for leg_name, leg_solo in (("materialized_default", False), ("materialized", True)):
          hj.set_option("solo", "1" if leg_solo else "0")
          mt = {"ms_join": [], "ms_close_gaps": [], "ms_total": []}
          for it in range(args.steps + 1):
            res = hj.phj(rk.data_ptr(), rv.data_ptr(), inner, sk.data_ptr(), sv.data_ptr(), outer, prm,
                         out=(jcols[0].ptr, jcols[1].ptr, jcols[2].ptr, cap, block),
                         stream=torch.cuda.current_stream().cuda_stream)
            stx = hj.stats()
            for k2 in mt:
                if it:
                    mt[k2].append(stx[k2])
          j = res[0]
          ok_rows = (list(res) == expect_local and hj.column_sums(jcols[0].ptr, j, 1, 1)[0] == expect_local[1]
                     and hj.column_sums(jcols[1].ptr, j, 1, 1)[0] == expect_local[2] and hj.column_sums(jcols[2].ptr, j, 1, 1)[0] == expect_local[3])
          mean = {k2: sum(v) / len(v) for k2, v in mt.items()}
          tj = mean["ms_join"] + mean["ms_close_gaps"]
          rw = 8 * n_tuples + 12 * j
          tj_each = [a_ + b_ for a_, b_ in zip(mt["ms_join"], mt["ms_close_gaps"])]
          out[leg_name] = {"rows": j, "steps": len(mt["ms_total"]), "statistic": "mean over the steps (min / max beside it)",
                               "row_stores": "plain (option solo)" if leg_solo else "non-temporal (the default policy)",
                               "ms_join": round(mean["ms_join"], 4), "ms_close_gaps": round(mean["ms_close_gaps"], 4),
                               "ms_total": round(mean["ms_total"], 4),
                               "ms_join_min_max": [round(min(mt["ms_join"]), 4), round(max(mt["ms_join"]), 4)],
                               "ms_total_min_max": [round(min(mt["ms_total"]), 4), round(max(mt["ms_total"]), 4)],
                               "gtuples_per_s": round(outer / mean["ms_total"] / 1e6, 2),
                               "join_phase_rw_GBs": round(rw / (tj * 1e-3) / 1e9, 1),
                               "join_phase_rw_frac": round(rw / (tj * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "join_phase_rw_frac_min_max": [round(rw / (max(tj_each) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                              round(rw / (min(tj_each) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)],
                               "roofline_join": roof(rw, tj, 1, stream_read_gbs),
                               "block_size": block, "rows_checksum_ok": bool(ok_rows)}
        hj.set_option("solo", "1" if solo else "0")
        out["materialized_default"]["traffic_source"] = attach_secondary_traffic(H, "materialized_default", [(out["materialized_default"]["roofline_join"], ("join_kernel", "mean"))])
        del jk, jo, ji
        for c in jcols:
            c.free()
    if rank == 0 and args.cpu_outer > 0:
        try:
            out["cpu_baseline"] = cpu_baseline(hj, args, args.algo, ("npj", "cpra") if extras else ())
            for which, entry in out["cpu_baseline"].pop("others", {}).items():          # the NPJ and CPRA legs' own CPU baselines
                out.get("secondary", {}).get(which, {})["cpu_baseline"] = entry
        except Exception as ex:          # the baseline is reported, never required
            out["cpu_baseline"] = {"value": None, "unit": "Gtuples/s", "cores": 0, "kind": "port",
                                   "sample": "failed: %r" % (ex,)}
        if extras:
            try:
                out["cpu_baseline_config0"] = cpu_baseline_config0(hj, args)
            except Exception as ex:
                out["cpu_baseline_config0"] = {"value": None, "unit": "Gtuples/s (probe side)", "cores": 0, "kind": "port",
                                               "sample": "failed: %r" % (ex,)}
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
    if comm is not None:
        comm.close()
    else:
        hj.close()
    if dist is not None:
        dist.destroy_process_group()
    if checksum_ok is False:
        sys.exit("join result does not match the analytic aggregates: got %r want %r" % (got, expect_global))


if __name__ == "__main__":
    main()
