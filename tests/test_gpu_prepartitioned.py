"""GPU tests of PHJ over relations that ARRIVE pass-1-partitioned (hjgpu_partition_packed_async,
hjgpu_phj_build_prepartitioned, hjgpu_phj_probe_prepartitioned_async): the receiving side of the multi-GPU CPRA, where
the exchange-level partitioning of every sender's own chunk IS pass 1 (cpra2.cpp:1757-1827, ownership 1868-1872, gather
1891-1959).  Here ONE GPU plays every sender and every receiver in turn; the host moves the "messages"."""
import numpy as np
import pytest

import hash_join_codes_knl_amd as H
from helpers import mulhi_hash, numpy_join

pytestmark = pytest.mark.gpu
FACTOR1 = 0x2C1B3C6D


def bounds(n, parts, alignment=16):
    part = (n // parts) & ~(alignment - 1)
    return [(part * t, n if t + 1 == parts else part * (t + 1)) for t in range(parts)]


def send(hj, keys, vals, chunks, fanout):
    """every sender partitions its own chunk: [(packed tuples, offsets[fanout + 1])] per chunk"""
    out = []
    for b, e in bounds(len(keys), chunks):
        n = e - b
        dk, dv = hj.column(np.concatenate([keys[b:e], np.zeros(1, np.uint32)])), hj.column(np.concatenate([vals[b:e], np.zeros(1, np.uint32)]))
        dt, do = hj.column(n + 16, np.uint64), hj.column(fanout + 1, np.uint64)
        hj.partition_packed_async(dk, dv, n, FACTOR1, fanout, dt, do)
        hj.synchronize()
        t, o = dt.download(n), do.download()
        # the operator's contract: partition p of the chunk = rows [o[p], o[p + 1]), every tuple in its partition
        assert o[0] == 0 and o[-1] == n and np.all(np.diff(o.astype(np.int64)) >= 0)
        p = mulhi_hash((t & np.uint64(0xFFFFFFFF)).astype(np.uint32), FACTOR1, fanout)
        assert np.array_equal(p, np.repeat(np.arange(fanout), np.diff(o.astype(np.int64))))
        assert np.array_equal(np.sort(t), np.sort((vals[b:e].astype(np.uint64) << np.uint64(32)) | keys[b:e].astype(np.uint64)))
        out.append((t, o.astype(np.int64)))
        for c in (dk, dv, dt, do):
            c.free()
    return out


def received(sent, g, k):
    """what rank g gets: from every chunk its partitions [g k, (g + 1) k), back to back; chunk offsets"""
    pieces = [t[o[g * k]:o[(g + 1) * k]] for t, o in sent]
    offs = np.concatenate([[0], np.cumsum([len(p) for p in pieces])])
    return (np.concatenate(pieces) if pieces else np.zeros(0, np.uint64)), offs


@pytest.mark.parametrize("ranks", [1, 2, 3, 8])
@pytest.mark.parametrize("kind", ["unique", "dups", "half", "tiny"])
def test_prepartitioned_relations_join_like_the_whole_relations(hj, oracle, ranks, kind):
    ik, iv, ok, ov = {"unique": lambda: oracle.generate(300_007, 61_003, seed=ranks),
                      "dups": lambda: oracle.generate(40_000, 250_000, seed=ranks),
                      "half": lambda: oracle.generate(200_000, 90_000, selectivity=0.5, seed=ranks),
                      "tiny": lambda: oracle.generate(37, 5, seed=ranks)}[kind]()
    want = numpy_join(ik, iv, ok, ov)
    k = max(1, 192 // ranks)
    fanout = ranks * k
    sent_r, sent_s = send(hj, ik, iv, ranks, fanout), send(hj, ok, ov, ranks, fanout)
    total = [0, 0, 0, 0]
    d_res = hj.column(4, np.uint64)
    for g in range(ranks):
        tr, offr = received(sent_r, g, k)
        ts, offs = received(sent_s, g, k)
        dr, ds = hj.column(np.concatenate([tr, np.zeros(2, np.uint64)]), np.uint64), hj.column(np.concatenate([ts, np.zeros(2, np.uint64)]), np.uint64)
        max_outer = max(len(ts), 1 << 16)
        hj.phj_build_prepartitioned(dr, hj.prepartitioned(FACTOR1, fanout, g * k, k, offr), max_outer)
        # the probe side in two batches cut at an arbitrary row (a piece of a piece is still sorted by partition)
        cut = (len(ts) * 3) // 7
        for lo, hi in ((0, cut), (cut, len(ts))):
            lay = hj.prepartitioned(FACTOR1, fanout, g * k, k, np.clip(offs, lo, hi))
            hj.phj_probe_prepartitioned_async(ds, lay, d_res)
            hj.get_async_status()
            total = [(a + int(b)) & ((1 << 64) - 1) for a, b in zip(total, d_res.download())]
        dr.free(); ds.free()
    d_res.free()
    assert tuple(total) == want, (ranks, kind)


def test_prepartitioned_layout_errors(hj, oracle):
    ik, iv, ok, ov = oracle.generate(5_000, 1_000, seed=1)
    sent = send(hj, ik, iv, 2, 8)
    t, offs = received(sent, 0, 4)
    d = hj.column(np.concatenate([t, np.zeros(2, np.uint64)]), np.uint64)
    d_res = hj.column(4, np.uint64)
    for bad in (hj.prepartitioned(FACTOR1 + 1, 8, 0, 4, offs),            # even factor
                hj.prepartitioned(FACTOR1, 8, 6, 4, offs),                # partitions beyond the fan-out
                hj.prepartitioned(FACTOR1, 8, 0, 4, offs[::-1])):         # offsets decrease
        with pytest.raises(H.HjGpuError) as e:
            hj.phj_build_prepartitioned(d, bad, 1 << 16)
        assert e.value.status == H.api.EINVAL
    hj.phj_build_prepartitioned(d, hj.prepartitioned(FACTOR1, 8, 0, 4, offs), 1 << 16)
    with pytest.raises(H.HjGpuError) as e:                                # another layout than the prepared build side's
        hj.phj_probe_prepartitioned_async(d, hj.prepartitioned(FACTOR1, 8, 4, 4, offs), d_res)
    assert e.value.status == H.api.EINVAL and "layout" in str(e.value)
    with pytest.raises(H.HjGpuError) as e:                                # batch beyond max_outer
        hj.phj_probe_prepartitioned_async(d, hj.prepartitioned(FACTOR1, 8, 0, 4, [0, 1 << 20, 1 << 21]), d_res)
    assert e.value.status == H.api.EINVAL
    # a plain prepared build side is not a pre-partitioned one
    rk, rv = hj.column(ik), hj.column(iv)
    hj.phj_build(rk, rv, len(ik), 1 << 16)
    with pytest.raises(H.HjGpuError) as e:
        hj.phj_probe_prepartitioned_async(d, hj.prepartitioned(FACTOR1, 8, 0, 4, offs), d_res)
    assert e.value.status == H.api.EINVAL
    for c in (d, d_res, rk, rv):
        c.free()


@pytest.mark.parametrize("n", [0, 1, 37, 70_001, 1_000_003])
@pytest.mark.parametrize("fanout,own_first,own_count", [(192, 48, 24), (192, 0, 96), (192, 168, 24), (8, 3, 1), (6, 0, 6), (64, 10, 0)])
def test_own_partitions_last(hj, oracle, n, fanout, own_first, own_count):
    """hjgpu_partition_packed_own_last_async: the partitions [own_first, own_first + own_count) end the output, all
    others keep their order in front of them; d_offsets stays the plain prefix of the counts (include/hjgpu.h gives a
    partition's first row from it).  Every tuple of the chunk is in its partition's rows, none is lost."""
    _, _, keys, vals = oracle.generate(max(n, 1), 16, seed=n % 97 + fanout)
    keys, vals = keys[:n], vals[:n]
    dk, dv = hj.column(np.concatenate([keys, np.zeros(1, np.uint32)])), hj.column(np.concatenate([vals, np.zeros(1, np.uint32)]))
    dt, do = hj.column(n + 16, np.uint64), hj.column(fanout + 1, np.uint64)
    hj.partition_packed_own_last_async(dk, dv, n, FACTOR1, fanout, own_first, own_count, dt, do)
    hj.synchronize()
    t, o = dt.download(n), do.download().astype(np.int64)
    assert o[0] == 0 and o[-1] == n and np.all(np.diff(o) >= 0)
    own_rows = o[own_first + own_count] - o[own_first]
    first = np.array([o[p] if p < own_first else (n - own_rows + o[p] - o[own_first] if p < own_first + own_count else o[p] - own_rows)
                      for p in range(fanout)], dtype=np.int64)
    want = np.empty(n, np.int64)
    for p in range(fanout):
        want[first[p]:first[p] + o[p + 1] - o[p]] = p
    got = mulhi_hash((t & np.uint64(0xFFFFFFFF)).astype(np.uint32), FACTOR1, fanout)
    assert np.array_equal(got, want)
    assert np.array_equal(np.sort(t), np.sort((vals.astype(np.uint64) << np.uint64(32)) | keys.astype(np.uint64)))
    # the counts are those of the plain operator
    hj.partition_packed_async(dk, dv, n, FACTOR1, fanout, dt, do)
    hj.synchronize()
    assert np.array_equal(do.download().astype(np.int64), o)
    for c in (dk, dv, dt, do):
        c.free()


def test_own_partitions_last_refuses_a_range_beyond_the_fanout(hj):
    dk, dv, dt, do = hj.column(64), hj.column(64), hj.column(80, np.uint64), hj.column(9, np.uint64)
    with pytest.raises(H.HjGpuError) as e:
        hj.partition_packed_own_last_async(dk, dv, 64, FACTOR1, 8, 6, 3, dt, do)
    assert e.value.status == H.api.EINVAL
    for c in (dk, dv, dt, do):
        c.free()


# ---- round 4: counts published with the partitions (cpra2.cpp:1783-1840) --------------------------------------------
def send_counted(hj, keys, vals, chunks, fanout, factor2, fanout2):
    """like send(), with the fused histogram of every chunk: counts[p1 * fanout2 + p2]"""
    out = []
    for b, e in bounds(len(keys), chunks):
        n = e - b
        dk, dv = hj.column(np.concatenate([keys[b:e], np.zeros(1, np.uint32)])), hj.column(np.concatenate([vals[b:e], np.zeros(1, np.uint32)]))
        dt, do, dc = hj.column(n + 16, np.uint64), hj.column(fanout + 1, np.uint64), hj.column(fanout * fanout2, np.uint64)
        hj.partition_packed_counted_async(dk, dv, n, FACTOR1, fanout, 0, 0, factor2, fanout2, dt, do, dc)
        hj.synchronize()
        t, o, cnt = dt.download(n), do.download().astype(np.int64), dc.download().astype(np.int64)
        k32 = (t & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        p1, p2 = mulhi_hash(k32, FACTOR1, fanout), mulhi_hash(k32, factor2, fanout2)
        # the operator's contract: the same partitions as the plain operator, and the fused histogram of the chunk
        assert np.array_equal(p1, np.repeat(np.arange(fanout), np.diff(o)))
        assert np.array_equal(cnt, np.bincount(p1.astype(np.int64) * fanout2 + p2, minlength=fanout * fanout2))
        out.append((t, o, cnt.reshape(fanout, fanout2)))
        for c in (dk, dv, dt, do, dc):
            c.free()
    return out


@pytest.mark.parametrize("ranks", [1, 3, 8])
@pytest.mark.parametrize("kind", ["unique", "dups", "tiny"])
def test_a_receiver_with_the_senders_counts_needs_no_histogram_pass(hj, oracle, ranks, kind):
    """hjgpu_partition_packed_counted_async + hjgpu_phj_probe_prepartitioned_counted_async: the receiver is handed the rows
    of every sender's fused histogram that describe its partitions (in the order of its pieces) and joins what arrived
    without counting it again; the result is the whole relations' join."""
    ik, iv, ok, ov = {"unique": lambda: oracle.generate(300_007, 61_003, seed=ranks),
                      "dups": lambda: oracle.generate(40_000, 250_000, seed=ranks),
                      "tiny": lambda: oracle.generate(37, 5, seed=ranks)}[kind]()
    want = numpy_join(ik, iv, ok, ov)
    k = max(1, 192 // ranks)
    fanout = ranks * k
    sent_r = send(hj, ik, iv, ranks, fanout)
    # every receiver plans the same second level: what the largest received build side needs
    most = max(sum(int(o[(g + 1) * k] - o[g * k]) for _, o in sent_r) for g in range(ranks))
    fanout2, factor2 = hj.prepartitioned_plan(most, k)
    assert fanout * fanout2 <= 32768
    sent_s = send_counted(hj, ok, ov, ranks, fanout, factor2, fanout2)
    total = [0, 0, 0, 0]
    d_res = hj.column(4, np.uint64)
    prm = H.PhjParams(fanout2=fanout2)
    for g in range(ranks):
        tr, offr = received(sent_r, g, k)
        ts, offs = received([(t, o) for t, o, _ in sent_s], g, k)
        counts = np.concatenate([c[g * k:(g + 1) * k].ravel() for _, _, c in sent_s]).astype(np.uint64)
        dr, ds = hj.column(np.concatenate([tr, np.zeros(2, np.uint64)]), np.uint64), hj.column(np.concatenate([ts, np.zeros(2, np.uint64)]), np.uint64)
        dc = hj.column(counts, np.uint64)
        hj.phj_build_prepartitioned(dr, hj.prepartitioned(FACTOR1, fanout, g * k, k, offr), max(len(ts), 1 << 16), prm)
        hj.phj_probe_prepartitioned_counted_async(ds, hj.prepartitioned(FACTOR1, fanout, g * k, k, offs), dc, d_res)
        hj.get_async_status()
        total = [(a + int(b)) & ((1 << 64) - 1) for a, b in zip(total, d_res.download())]
        for c in (dr, ds, dc):
            c.free()
    d_res.free()
    assert tuple(total) == want, (ranks, kind)


def test_counted_partitions_refuse_what_the_lds_histogram_cannot_hold(hj):
    dk, dv, dt, do, dc = hj.column(64), hj.column(64), hj.column(80, np.uint64), hj.column(193, np.uint64), hj.column(16, np.uint64)
    for factor2, fanout2 in ((0x85EBCA6B, 171), (FACTOR1, 4), (0x85EBCA6A, 4)):      # 192 x 171 > 32768; the same factor; an even factor
        with pytest.raises(H.HjGpuError) as e:
            hj.partition_packed_counted_async(dk, dv, 64, FACTOR1, 192, 0, 0, factor2, fanout2, dt, do, dc)
        assert e.value.status == H.api.EINVAL
    for c in (dk, dv, dt, do, dc):
        c.free()


def test_partitioning_keeps_every_tuple_while_joins_run_on_another_stream(hj):
    """The condition under which a K6 pass-1 instance WITH a private segment lost ~1e-5 of its stores (DESIGN section 3,
    profiles/r04_scratch_repro.txt: tools/scratch_two_streams.py, 23 of 40 steps wrong with one private word, 0 alone):
    two independent contexts on one device, one partitioning on its stream while the other runs whole joins on a
    second stream.  The shipped library has no private segment (tests/test_kernel_resources.py) and must lose
    nothing: every output's 32-bit word sum equals the input's and no slot stays unwritten."""
    import torch                                            # (initialised by the session's context fixture, before any library call)
    a, b = H.HjGpu(0), H.HjGpu(0)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    inner, outer, n, fanout = 16_000_000, 256_000_000, 48_000_000, 192
    ik, iv, ok, ov = b.column(inner), b.column(inner), b.column(outer), b.column(outer)
    b.generate(1, inner, outer, 0, outer, 0x2545F491, 0x9E3779B1, ik, iv, ok, ov)
    want_join = [outer, *b.column_sums(ok, outer, 0x9E3779B1, 0x2545F491)]
    want_words = a.column_sums(ok, n, 1, 1)[0] + a.column_sums(ov, n, 1, 1)[0]
    d_res = b.column(4, np.uint64)
    off = a.column(fanout + 1, np.uint64)
    out = torch.zeros(n + 64, dtype=torch.int64, device="cuda:0")       # zeroed before every call: a lost store stays 0
    torch.cuda.synchronize()
    for step in range(12):
        for _ in range(2):
            b.phj_async(ik, iv, inner, ok, ov, outer, None, d_res, sb.cuda_stream)
        for _ in range(3):
            with torch.cuda.stream(sa):
                out.zero_()
            a.partition_packed_async(ok, ov, n, FACTOR1, fanout, out.data_ptr(), off, sa.cuda_stream)
        got_words = a.column_sums(out.data_ptr(), 2 * n, 1, 1, sa.cuda_stream)[0]
        a.synchronize(sa.cuda_stream)
        b.synchronize(sb.cuda_stream)
        assert got_words == want_words, f"step {step}: the partitioned relation lost or changed tuples"
        assert [int(x) for x in d_res.download()] == want_join, f"step {step}: the neighbouring join"
        o = off.download().astype(np.int64)
        assert o[0] == 0 and o[-1] == n and np.all(np.diff(o) >= 0)
    for c in (ik, iv, ok, ov, d_res, off):
        c.free()
    a.close()
    b.close()
