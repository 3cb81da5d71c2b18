"""GPU tests of the grouped plans (a third partitioning pass: hjgpu_api.hip phj_grouped; the reference plans up to four
passes from the partition count, phj.cpp:1791-1808).  The options "group_from" / "group_inner" bring the plan down to
test sizes: both relations are split into key-disjoint groups by pass 0, every group is joined by the two-pass plan,
and aggregates and rows must equal the oracle's / the definition's - the same bar as the plain plan (bit-exact)."""
import numpy as np
import pytest

import hash_join_codes_knl_amd as H
from helpers import numpy_join, materialised_rows, sort_rows

pytestmark = pytest.mark.gpu


@pytest.fixture()
def grouped(hj):
    """a context whose plans are grouped from 1000 build tuples on (after the session's context: torch, where it is
    installed, has to initialise the GPU first - conftest.py)"""
    ctx = H.HjGpu()
    ctx.set_option("group_from", "1000")
    ctx.set_option("group_always", "1")
    yield ctx
    ctx.close()


def _cols(hj, *arrays):
    return [hj.column(a) for a in arrays]


def _free(*cols):
    for c in cols:
        c.free()


SHAPES = {
    # name: (outer, inner, selectivity, seed)
    "unique": (400_000, 120_000, 1.0, 11),
    "build_dups": (30_000, 240_000, 1.0, 12),           # 8 copies of every build key
    "half": (250_000, 90_000, 0.5, 13),
    "none": (50_000, 20_000, 0.0, 14),
    "probe_small": (900, 60_000, 1.0, 15),
}


@pytest.mark.parametrize("groups", [2, 7, 64, 192])
@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("algo", ["phj", "cpra"])
def test_grouped_aggregates_equal_the_definition(grouped, oracle, algo, shape, groups):
    outer, inner, sel, seed = SHAPES[shape]
    ik, iv, ok, ov = oracle.generate(outer, inner, selectivity=sel, seed=seed)
    want = numpy_join(ik, iv, ok, ov)
    per = -(-len(ik) // groups)
    groups = -(-len(ik) // per)
    grouped.set_option("group_inner", str(per))
    rk, rv, sk, sv = _cols(grouped, ik, iv, ok, ov)
    got = getattr(grouped, algo)(rk, rv, len(ik), sk, sv, len(ok))
    st = grouped.stats()
    _free(rk, rv, sk, sv)
    assert got == want
    assert st["groups"] == groups and st["ms_scatter0"] > 0 and st["ms_total"] >= st["ms_scatter0"] + st["ms_join"]


def test_the_plain_plan_reports_no_groups(hj, oracle):
    ik, iv, ok, ov = oracle.generate(20_000, 5_000, seed=3)
    rk, rv, sk, sv = _cols(hj, ik, iv, ok, ov)
    assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == numpy_join(ik, iv, ok, ov)
    st = hj.stats()
    _free(rk, rv, sk, sv)
    assert st["groups"] == 0 and st["ms_scatter0"] == 0


@pytest.mark.parametrize("groups", [3, 40])
@pytest.mark.parametrize("shape", ["unique", "build_dups", "half"])
def test_grouped_rows_equal_the_definition(grouped, oracle, shape, groups):
    outer, inner, sel, seed = SHAPES[shape]
    ik, iv, ok, ov = oracle.generate(outer, inner, selectivity=sel, seed=seed)
    want = numpy_join(ik, iv, ok, ov)
    wk, wo, wi = materialised_rows(ik, iv, ok, ov)
    grouped.set_option("group_inner", str(-(-len(ik) // groups)))
    block = 256
    capacity = grouped.output_capacity(1, len(ok), want[0], block)
    rk, rv, sk, sv = _cols(grouped, ik, iv, ok, ov)
    jk, jo, ji = grouped.column(capacity), grouped.column(capacity), grouped.column(capacity)
    got = grouped.phj(rk, rv, len(ik), sk, sv, len(ok), None, out=(jk, jo, ji, capacity, block))
    assert got == want
    n = got[0]
    gk, go, gi = sort_rows(jk.download(n), jo.download(n), ji.download(n))
    _free(rk, rv, sk, sv, jk, jo, ji)
    assert np.array_equal(gk, wk) and np.array_equal(go, wo) and np.array_equal(gi, wi)


def test_grouped_rows_that_do_not_fit_are_reported_with_exact_counts(grouped, oracle):
    ik, iv, ok, ov = oracle.generate(200_000, 50_000, seed=21)
    want = numpy_join(ik, iv, ok, ov)
    grouped.set_option("group_inner", "10000")
    rk, rv, sk, sv = _cols(grouped, ik, iv, ok, ov)
    cap = 256 * 300                                        # room for the first group or two only
    jk, jo, ji = grouped.column(cap), grouped.column(cap), grouped.column(cap)
    with pytest.raises(H.HjGpuError) as e:
        grouped.phj(rk, rv, len(ik), sk, sv, len(ok), None, out=(jk, jo, ji, cap, 256))
    assert e.value.status == H.api.EOVERFLOW
    # the same call through the asynchronous form: the device-side result holds the exact aggregates
    d_res = grouped.column(4, np.uint64)
    grouped.set_async_output((jk, jo, ji, cap, 256))
    grouped.phj_async(rk, rv, len(ik), sk, sv, len(ok), None, d_res)
    with pytest.raises(H.HjGpuError) as e:
        grouped.get_async_status()
    assert e.value.status == H.api.EOVERFLOW
    assert tuple(int(x) for x in d_res.download()) == want
    _free(rk, rv, sk, sv, jk, jo, ji, d_res)


def test_grouped_first_match_only(grouped, oracle):
    """HJGPU_FLAG_UNIQUE (-D_UNIQUE, phj.cpp:635-637): the groups are key-disjoint, so first-match-only inside every
    group is first-match-only of the whole join: every probe tuple with a partner counts once"""
    ik, iv, ok, ov = oracle.generate(40_000, 320_000, seed=31)          # 8 copies per build key
    grouped.set_option("group_inner", "40000")
    rk, rv, sk, sv = _cols(grouped, ik, iv, ok, ov)
    count, sum_keys, sum_outer, _ = grouped.phj(rk, rv, len(ik), sk, sv, len(ok), H.PhjParams(flags=H.api.FLAG_UNIQUE))
    _free(rk, rv, sk, sv)
    hit = np.isin(ok, ik)
    assert count == int(hit.sum())
    assert sum_keys == int(ok[hit].astype(np.uint64).sum()) and sum_outer == int(ov[hit].astype(np.uint64).sum())


def test_an_explicit_plan_is_never_grouped(grouped, oracle):
    ik, iv, ok, ov = oracle.generate(60_000, 20_000, seed=41)
    grouped.set_option("group_inner", "5000")
    rk, rv, sk, sv = _cols(grouped, ik, iv, ok, ov)
    assert grouped.phj(rk, rv, len(ik), sk, sv, len(ok), H.PhjParams(fanout1=8, fanout2=4)) == numpy_join(ik, iv, ok, ov)
    st = grouped.stats()
    _free(rk, rv, sk, sv)
    assert st["groups"] == 0 and (st["fanout1"], st["fanout2"]) == (8, 4)


def test_grouped_plan_at_a_size_it_is_chosen_for_by_default(hj):
    """|R| = 900 M unique build keys (four table fills per partition in two passes), |S| = 2.6 G: the default options
    group this join (15 groups of about 64 M build tuples); aggregates are analytic (selectivity 1: every probe tuple matches once).  The same build
    side with |S| = 500 M stays with the two-pass plan: the extra pass would cost more than the fills it saves."""
    hj = H.HjGpu()                                         # its own context: the workspace goes away with it
    inner, outer = 900_000_000, 2_600_000_000
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(7, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)
    got = hj.phj(ik, iv, inner, ok, ov, outer)
    st = hj.stats()
    small = 500_000_000
    sums_small = hj.column_sums(ok, small, 0x9E3779B1, 0x2545F491)
    got_small = hj.phj(ik, iv, inner, ok, ov, small)
    st_small = hj.stats()
    _free(ik, iv, ok, ov)
    hj.close()
    assert got == (outer, sums[0], sums[1], sums[2]) and st["groups"] == 15
    assert got_small == (small, sums_small[0], sums_small[1], sums_small[2]) and st_small["groups"] == 0


@pytest.mark.parametrize("algo", ["phj", "cpra"])
def test_grouped_plan_through_the_enqueue_only_form_returns_at_once(grouped, algo):
    """hjgpu_phj_async / hjgpu_cpra_async with a grouped plan: the groups are planned ON THE DEVICE, so the call is enqueue-only like any
    other join - it returns before the join has run, the result is in d_result once the caller's stream is idle, calls queue up back to
    back - and the blocking form and the host-planned form (option group_device = 0) give the same aggregates."""
    import time
    inner, outer = 20_000_000, 100_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    grouped.set_option("group_inner", "2500000")          # 8 groups
    ik, iv, ok, ov = (grouped.column(n) for n in (inner, inner, outer, outer))
    grouped.generate(7, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    sums = grouped.column_sums(ok, outer, fo, fi)
    want = (outer, sums[0], sums[1], sums[2])
    blocking = getattr(grouped, algo)(ik, iv, inner, ok, ov, outer)
    assert blocking == want and grouped.stats()["groups"] == 8
    t0 = time.perf_counter()
    getattr(grouped, algo)(ik, iv, inner, ok, ov, outer)
    t_blocking = time.perf_counter() - t0
    d = [grouped.column(4, np.uint64) for _ in range(2)]
    getattr(grouped, algo + "_async")(ik, iv, inner, ok, ov, outer, None, d[0])
    grouped.synchronize()
    assert grouped.get_async_status() is None
    # the stream (the legacy one: torch's default stream too) is given ~40 ms of other work first: a call that waited for the device
    # would take that long, and the stream is still busy when an enqueue-only call returns
    import torch
    junk = torch.zeros(1 << 29, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    for _ in range(60):
        junk.add_(1)
    t0 = time.perf_counter()
    getattr(grouped, algo + "_async")(ik, iv, inner, ok, ov, outer, None, d[0])
    t_call = time.perf_counter() - t0
    busy = not torch.cuda.default_stream().query()
    getattr(grouped, algo + "_async")(ik, iv, inner, ok, ov, outer, None, d[1])        # queues behind the first
    grouped.synchronize()
    grouped.get_async_status()
    for r in d:
        assert tuple(int(x) for x in r.download()) == want
    st = grouped.stats()
    assert st["groups"] == 8 and st["ms_scatter0"] > 0 and st["ms_join"] > 0 and st["ms_total"] >= st["ms_scatter0"] + st["ms_join"]
    assert busy and t_call < 0.02, (busy, t_call, t_blocking)
    del junk
    grouped.set_option("group_device", "0")               # the host-planned form: the call itself waits for pass 0 and for every group
    getattr(grouped, algo + "_async")(ik, iv, inner, ok, ov, outer, None, d[0])
    grouped.synchronize()
    assert tuple(int(x) for x in d[0].download()) == want
    assert getattr(grouped, algo)(ik, iv, inner, ok, ov, outer) == want and grouped.stats()["groups"] == 8
    grouped.set_option("group_device", "1")
    _free(ik, iv, ok, ov, *d)


def test_a_group_beyond_the_planned_workspace_marks_the_result_and_is_joined_again(hj, oracle):
    """Device-planned groups have workspace for (1 + group_slack / 100) x the mean group.  With few distinct build keys the largest group is far
    above that: the enqueue-only join skips it, d_result then holds all ones (never a plausible partial count), and hjgpu_get_async_status
    does the join again host-planned and leaves the right aggregates (and rows) where the caller expects them; the blocking form does the
    same inside the call."""
    rng = np.random.default_rng(11)
    distinct = rng.choice(np.arange(1, 1 << 31, dtype=np.uint32), size=24, replace=False)
    ik = np.repeat(distinct, 400_000)                     # 24 keys in 12 groups of 800 000 rows on average: a group with three of them has 1.2 M
    iv = rng.integers(0, 1 << 32, size=len(ik), dtype=np.uint32)
    ok = rng.choice(distinct, size=3000)
    ov = rng.integers(0, 1 << 32, size=len(ok), dtype=np.uint32)
    # (every probe row matches 400 000 build rows: the aggregates in closed form)
    per_key_iv = {int(k): int(iv[i * 400_000:(i + 1) * 400_000].sum(dtype=np.uint64)) for i, k in enumerate(distinct)}
    M = (1 << 64) - 1
    def closed_form(keys, vals):
        return (len(keys) * 400_000, int(keys.astype(np.uint64).sum()) * 400_000 & M, int(vals.astype(np.uint64).sum()) * 400_000 & M,
                sum(per_key_iv[int(k)] for k in keys) & M)
    want = closed_form(ok, ov)
    ctx = H.HjGpu()
    try:
        for n, v in (("group_from", "1000"), ("group_always", "1"), ("group_inner", str(len(ik) // 12)), ("group_slack", "10")):
            ctx.set_option(n, v)
        rk, rv, sk, sv = _cols(ctx, ik, iv, ok, ov)
        d = ctx.column(4, np.uint64)
        ctx.phj_async(rk, rv, len(ik), sk, sv, len(ok), None, d)
        ctx.synchronize()
        assert tuple(int(x) for x in d.download()) == ((1 << 64) - 1,) * 4          # a group was skipped: the result is MARKED, not partial
        ctx.get_async_status()                                                       # ... and joined again, host-planned
        assert tuple(int(x) for x in d.download()) == want
        assert ctx.phj(rk, rv, len(ik), sk, sv, len(ok)) == want                     # the blocking form falls back inside the call
        assert ctx.cpra(rk, rv, len(ik), sk, sv, len(ok)) == want
        # rows: the marked join's rows are not valid; after the status call they are the join's
        ok2, ov2 = ok[:150].copy(), ov[:150].copy()
        want2 = closed_form(ok2, ov2)
        sk2, sv2 = ctx.column(ok2), ctx.column(ov2)
        cap = ctx.output_capacity(1, len(ok2), want2[0], 4096)
        cols = [ctx.column(cap) for _ in range(3)]
        ctx.set_async_output((cols[0], cols[1], cols[2], cap, 4096))
        ctx.phj_async(rk, rv, len(ik), sk2, sv2, len(ok2), None, d)
        ctx.get_async_status()
        assert tuple(int(x) for x in d.download()) == want2
        rows = [c.download()[:want2[0]] for c in cols]
        assert tuple(int(r.sum(dtype=np.uint64)) & M for r in rows) == want2[1:]
        _free(sk2, sv2)
        _free(rk, rv, sk, sv, d, *cols)
    finally:
        ctx.close()


def test_first_grouped_join_of_a_context_through_the_enqueue_only_form(hj, oracle):
    """A fresh context whose first join is an enqueue-only grouped one (the workspace grows inside the call), with few distinct build keys,
    so that the largest group is far above the mean - beyond the device plan's workspace, i.e. hjgpu_get_async_status joins again
    host-planned; then a larger join on the same context (everything grows again)."""
    rng = np.random.default_rng(5)
    for scale in (1, 6):
        ctx = H.HjGpu()
        try:
            ctx.set_option("group_from", "1000")
            ctx.set_option("group_always", "1")
            for rep in (1, scale):
                distinct = rng.choice(np.arange(1, 1 << 31, dtype=np.uint32), size=64, replace=False)
                ik = np.repeat(distinct, 500 * rep)
                iv = rng.integers(0, 1 << 32, size=len(ik), dtype=np.uint32)
                ok = rng.choice(distinct, size=50_000 * rep)
                ov = rng.integers(0, 1 << 32, size=len(ok), dtype=np.uint32)
                want = numpy_join(ik, iv, ok, ov)
                ctx.set_option("group_inner", str(len(ik) // 8))
                rk, rv, sk, sv = _cols(ctx, ik, iv, ok, ov)
                d = ctx.column(4, np.uint64)
                ctx.phj_async(rk, rv, len(ik), sk, sv, len(ok), None, d)
                ctx.synchronize()
                ctx.get_async_status()
                assert tuple(int(x) for x in d.download()) == want
                assert ctx.stats()["groups"] >= 8
                _free(rk, rv, sk, sv, d)
        finally:
            ctx.close()


def test_asynchronous_grouped_joins_in_a_process_with_more_streams_than_hardware_queues(hj, oracle):
    """A grouped join is one stream-ordered sequence on the caller's stream (planned on the device: no worker thread, no stream held in
    hardware): in a process with far more streams than hardware queues - 24 busy ones here - joins enqueued on several of them in turn
    (and on the legacy stream) neither stall each other nor the other streams."""
    import torch
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(device=dev) for _ in range(24)]
    junk = torch.zeros(1 << 20, dtype=torch.int32, device=dev)
    for st in streams:                                   # every stream has run something: its hardware queue is assigned
        with torch.cuda.stream(st):
            junk.add_(1)
    torch.cuda.synchronize()
    ik, iv, ok, ov = oracle.generate(400_000, 120_000, seed=21)
    want = numpy_join(ik, iv, ok, ov)
    # one context per stream (a context owns one workspace: include/hjgpu.h), six joins in flight on six streams at once
    ctxs = [H.HjGpu() for _ in range(6)]
    try:
        for ctx in ctxs:
            ctx.set_option("group_from", "1000")
            ctx.set_option("group_always", "1")
            ctx.set_option("group_inner", str(len(ik) // 6))
        rk, rv, sk, sv = _cols(ctxs[0], ik, iv, ok, ov)
        d = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(6)]
        torch.cuda.synchronize()
        callers = [streams[0].cuda_stream, streams[5].cuda_stream, streams[11].cuda_stream, None, streams[23].cuda_stream, streams[2].cuda_stream]
        for rounds in range(3):
            for ctx, res, st in zip(ctxs, d, callers):
                ctx.phj_async(rk, rv, len(ik), sk, sv, len(ok), None, res.data_ptr(), st)
                for other in streams[::3]:               # the other streams keep working beside the joins
                    with torch.cuda.stream(other):
                        junk.add_(1)
        for ctx, st in zip(ctxs, callers):
            ctx.get_async_status(st)
        torch.cuda.synchronize()
        for res in d:
            assert tuple(int(x) & ((1 << 64) - 1) for x in res.tolist()) == want
        assert all(ctx.stats()["groups"] >= 6 for ctx in ctxs)
        _free(rk, rv, sk, sv)
    finally:
        for ctx in ctxs:
            ctx.close()


def test_pass_zero_is_independent_of_the_callers_pass_factors(grouped, oracle):
    """A caller's factor1 (or factor2) equal to the library's default pass-0 multiplier must not line the groups' own first pass
    up with pass 0 (every key of a group would land in 1 / G of its pass-1 partitions: correct through multi-fill tables, but
    what the grouped plan exists to avoid): pass 0 then takes another multiplier.  Results equal the definition either way."""
    ik, iv, ok, ov = oracle.generate(300_000, 100_000, seed=31)
    want = numpy_join(ik, iv, ok, ov)
    grouped.set_option("group_inner", "12500")            # 8 groups
    rk, rv, sk, sv = _cols(grouped, ik, iv, ok, ov)
    for prm in (H.PhjParams(factor1=0x7FEB352D), H.PhjParams(factor2=0x7FEB352D), H.PhjParams(factor1=0x7FEB352D, factor2=0x846CA68B)):
        assert grouped.phj(rk, rv, len(ik), sk, sv, len(ok), prm) == want
        assert grouped.stats()["groups"] == 8
    _free(rk, rv, sk, sv)
