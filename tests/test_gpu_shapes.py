"""GPU tests of workload shapes away from the benchmark's: tiny build side against
a large probe side (slices), build side larger than probe side, heavy skew
(multi-fill overflow + chained fallback), selectivity 0, many duplicates on both sides."""
import os

import numpy as np
import pytest

import hash_join_codes_knl_amd as H
from helpers import numpy_join, materialised_rows, sort_rows

pytestmark = pytest.mark.gpu


def _run(hj, ik, iv, ok, ov, algos=("npj", "phj", "cpra"), prm=None):
    rk, rv, sk, sv = hj.column(ik), hj.column(iv), hj.column(ok), hj.column(ov)
    out = {a: getattr(hj, a)(rk, rv, len(ik), sk, sv, len(ok), prm if a != "npj" else None) for a in algos}
    for c in (rk, rv, sk, sv):
        c.free()
    return out


def test_tiny_build_huge_probe_uses_slices(hj):
    rng = np.random.default_rng(1)
    ik = np.arange(1, 301, dtype=np.uint32) * np.uint32(2654435761)
    iv = ik ^ np.uint32(0x55AA55AA)
    ok = ik[rng.integers(0, len(ik), size=20_000_000)]
    ov = (ok.astype(np.uint64) * 7 + 1).astype(np.uint32)
    want = numpy_join(ik, iv, ok, ov)
    assert want[0] == len(ok)
    for a, got in _run(hj, ik, iv, ok, ov).items():
        assert got == want, a


def test_build_larger_than_probe(hj):
    rng = np.random.default_rng(2)
    ik = np.unique(rng.integers(1, 2**32, size=6_000_000, dtype=np.uint64).astype(np.uint32))
    iv = ik * np.uint32(3)
    ok = np.concatenate([ik[rng.integers(0, len(ik), size=150_000)],
                         rng.integers(1, 2**32, size=150_000, dtype=np.uint64).astype(np.uint32)])
    ov = ok * np.uint32(5)
    want = numpy_join(ik, iv, ok, ov)
    for a, got in _run(hj, ik, iv, ok, ov).items():
        assert got == want, a


def test_zipf_heavy_hitters_both_sides(hj):
    """A few keys carry most tuples on both sides: partitions far beyond one LDS fill
    (multi-fill path) and >2 copies per key (cuckoo build fails -> chained fallback)."""
    rng = np.random.default_rng(3)
    base = np.unique(rng.integers(1, 2**32, size=3000, dtype=np.uint64).astype(np.uint32))
    w = 1.0 / np.arange(1, len(base) + 1) ** 1.2
    w /= w.sum()
    ik = base[rng.choice(len(base), size=400_000, p=w)]
    ok = base[rng.choice(len(base), size=300_000, p=w)]
    iv = (ik.astype(np.uint64) * 11 + np.arange(len(ik), dtype=np.uint64)).astype(np.uint32)
    ov = (ok.astype(np.uint64) * 13 + np.arange(len(ok), dtype=np.uint64)).astype(np.uint32)
    want = numpy_join(ik, iv, ok, ov)
    assert want[0] > 50_000_000            # heavy output
    for prm in (None, H.PhjParams(fanout1=4, fanout2=1), H.PhjParams(fanout1=64, fanout2=64, chunks=5)):
        for a, got in _run(hj, ik, iv, ok, ov, algos=("phj", "cpra"), prm=prm).items():
            assert got == want, (a, prm and (prm.fanout1, prm.fanout2))
    assert _run(hj, ik, iv, ok, ov, algos=("npj",))["npj"] == want


def test_exactly_two_copies_per_build_key_stay_on_cuckoo_path(hj):
    rng = np.random.default_rng(4)
    u = np.unique(rng.integers(1, 2**32, size=200_000, dtype=np.uint64).astype(np.uint32))
    ik = np.concatenate([u, u]); rng.shuffle(ik)
    iv = np.arange(len(ik), dtype=np.uint32)
    ok = u[rng.integers(0, len(u), size=1_000_000)]
    ov = ok ^ np.uint32(0xDEADBEEF)
    want = numpy_join(ik, iv, ok, ov)
    assert want[0] == 2 * len(ok)
    for a, got in _run(hj, ik, iv, ok, ov).items():
        assert got == want, a


def test_no_matches_at_all(hj):
    rng = np.random.default_rng(5)
    ik = (rng.integers(1, 2**31, size=500_000, dtype=np.uint64) * 2).astype(np.uint32)        # even keys
    ok = (rng.integers(0, 2**31, size=2_000_000, dtype=np.uint64) * 2 + 1).astype(np.uint32)  # odd keys
    for a, got in _run(hj, ik, ik, ok, ok).items():
        assert got == (0, 0, 0, 0), a


def test_materialised_heavy_output(hj):
    """Output far larger than both inputs (16 x 16 duplicates): block claiming + close_gaps."""
    rng = np.random.default_rng(6)
    u = np.unique(rng.integers(1, 2**32, size=4000, dtype=np.uint64).astype(np.uint32))
    ik = np.repeat(u, 16); ok = np.repeat(u, 16)
    iv = np.arange(len(ik), dtype=np.uint32); ov = np.arange(len(ok), dtype=np.uint32) * np.uint32(3)
    want = numpy_join(ik, iv, ok, ov)
    assert want[0] == len(u) * 256
    block, cap = 256, (want[0] // 256 + 8200) * 256
    rk, rv, sk, sv = hj.column(ik), hj.column(iv), hj.column(ok), hj.column(ov)
    jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
    for algo in ("phj", "npj"):
        res = getattr(hj, algo)(rk, rv, len(ik), sk, sv, len(ok), None, out=(jk, jo, ji, cap, block))
        assert res == want
        k, o, i = jk.download(res[0]), jo.download(res[0]), ji.download(res[0])
        assert int(k.astype(np.uint64).sum()) == want[1] and int(o.astype(np.uint64).sum()) == want[2]
        assert int(i.astype(np.uint64).sum()) == want[3]
        # every row is a real (outer, inner) pair with equal keys
        assert np.array_equal(ik[i], k) and np.array_equal(ok[(o.astype(np.uint64) * pow(3, -1, 2**32) % 2**32).astype(np.int64)], k)
    for c in (rk, rv, sk, sv, jk, jo, ji):
        c.free()


def test_device_generator_zipf_skews_and_keeps_the_contract(hj):
    """hjgpu_generate_zipf: same statistical contract as the uniform generator (unique non-zero
    build keys, every build key at least once on the probe side, payload = key * factor), but
    the repeat picks follow a Zipf law: the hottest key must carry far more than the mean."""
    inner, outer = 50_000, 2_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate_zipf(7, inner, outer, 0, inner, 0, outer, fi, fo, 1.0, ik, iv, ok, ov)
    hik, hiv, hok, hov = ik.download(), iv.download(), ok.download(), ov.download()
    assert len(np.unique(hik)) == inner and (hik != 0).all()
    assert np.array_equal(hiv, hik * np.uint32(fi)) and np.array_equal(hov, hok * np.uint32(fo))
    keys, counts = np.unique(hok, return_counts=True)
    assert np.array_equal(keys, np.sort(hik))                 # every build key at least once
    assert counts.max() > 50 * outer / inner                  # Zipf(1): top key ~ 1/ln(N) of the picks
    # shards are independent: two halves generated separately equal the whole
    h1, h2 = hj.column(outer // 2), hj.column(outer - outer // 2)
    v1, v2 = hj.column(outer // 2), hj.column(outer - outer // 2)
    hj.generate_zipf(7, inner, outer, 0, 0, 0, outer // 2, fi, fo, 1.0, None, None, h1, v1)
    hj.generate_zipf(7, inner, outer, 0, 0, outer // 2, outer - outer // 2, fi, fo, 1.0, None, None, h2, v2)
    assert np.array_equal(np.concatenate([h1.download(), h2.download()]), hok)
    want = numpy_join(hik, hiv, hok, hov)
    assert want[0] == outer
    for a in ("npj", "phj", "cpra"):
        assert getattr(hj, a)(ik, iv, inner, ok, ov, outer) == want, a
    for c in (ik, iv, ok, ov, h1, h2, v1, v2):
        c.free()


@pytest.mark.parametrize("zipf", [0.75, 1.25])
def test_full_size_zipf_probe_side(hj, zipf):
    """|R| = 64 M, |S| = 512 M with a Zipf probe side: one partition receives tens of millions of
    probe tuples (its work items are slices), the aggregates stay the column sums of S."""
    inner, outer = 64_000_000, 512_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate_zipf(3, inner, outer, 0, inner, 0, outer, fi, fo, zipf, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, fo, fi)
    want = (outer, sums[0], sums[1], sums[2])
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    assert hj.cpra(ik, iv, inner, ok, ov, outer, H.PhjParams(chunks=4)) == want
    assert hj.npj(ik, iv, inner, ok, ov, outer) == want
    for c in (ik, iv, ok, ov):
        c.free()


def test_one_build_key_with_a_million_copies(hj):
    """A build partition hundreds of times the LDS table (one key 1 M times among 200 K unique keys), probed by
    20 K copies of that key: the table of that partition is filled ~250 times and its probe slice re-streamed
    (J = 2 x 10^10 + the unique matches), all three algorithms, default plans and a forced tiny fan-out."""
    rng = np.random.default_rng(99)
    u = np.unique(rng.integers(1, 2**32, size=200_000, dtype=np.uint64).astype(np.uint32))
    hot = u[1234]
    ik = np.concatenate([u, np.full(1_000_000, hot, np.uint32)])
    rng.shuffle(ik)
    iv = rng.integers(0, 2**32, size=len(ik), dtype=np.uint64).astype(np.uint32)
    ok = np.concatenate([u[rng.integers(0, len(u), size=500_000)], np.full(20_000, hot, np.uint32)])
    rng.shuffle(ok)
    ov = rng.integers(0, 2**32, size=len(ok), dtype=np.uint64).astype(np.uint32)
    want = numpy_join(ik, iv, ok, ov)
    assert want[0] > 2 * 10**10
    rk, rv, sk, sv = (hj.column(c) for c in (ik, iv, ok, ov))
    assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == want
    assert hj.phj(rk, rv, len(ik), sk, sv, len(ok), H.PhjParams(fanout1=3, fanout2=2)) == want
    assert hj.cpra(rk, rv, len(ik), sk, sv, len(ok), H.PhjParams(chunks=5)) == want
    for c in (rk, rv, sk, sv):
        c.free()


@pytest.mark.parametrize("inner", [1, 17, 4095, 4096, 4097, 6963, 6964])
def test_broadcast_join_of_a_build_side_that_fits_one_table(hj, inner):
    """Build sides of at most one LDS table (4096 rows in the 8 K-slot table, 6963 in the 16 K-slot one) take the
    broadcast path: nothing is partitioned, every slice of the caller's probe columns builds the table and
    probes.  Around the thresholds, with duplicates,
    key 0, misses, probe columns that start inside an allocation, rows materialised; HJGPU_NO_BROADCAST=1
    (the partitioned plan) must agree."""
    rng = np.random.default_rng(1000 + inner)
    base = np.unique(np.concatenate([[0], rng.integers(0, 2**32, size=max(1, inner // 2), dtype=np.uint64)]).astype(np.uint32))
    ik = base[rng.integers(0, len(base), size=inner)]
    iv = rng.integers(0, 2**32, size=inner, dtype=np.uint64).astype(np.uint32)
    outer = 1_300_003
    ok = np.where(rng.random(outer) < 0.6, base[rng.integers(0, len(base), size=outer)],
                  rng.integers(0, 2**32, size=outer, dtype=np.uint64).astype(np.uint32)).astype(np.uint32)
    ov = rng.integers(0, 2**32, size=outer, dtype=np.uint64).astype(np.uint32)
    rk, rv = hj.column(ik), hj.column(iv)
    pad = 4                                             # probe columns start 16 bytes into their allocations
    sk_all, sv_all = hj.column(np.concatenate([np.zeros(pad, np.uint32), ok])), hj.column(np.concatenate([np.zeros(pad, np.uint32), ov]))
    sk, sv = sk_all.ptr + 4 * pad, sv_all.ptr + 4 * pad
    want = numpy_join(ik, iv, ok, ov)
    assert hj.phj(rk, rv, inner, sk, sv, outer) == want
    st = hj.stats()
    assert ((st["fanout1"], st["fanout2"]) == (1, 1)) == (inner <= 6963)
    hj.set_option("no_broadcast", 1)
    try:
        assert hj.phj(rk, rv, inner, sk, sv, outer) == want
        assert hj.stats()["fanout1"] >= 2
    finally:
        hj.set_option("no_broadcast", 0)
    if want[0] <= 20_000_000:
        block = 1024
        cap = (want[0] // block + hj.device_info()["compute_units"] * 16 + 8) * block
        jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
        assert hj.phj(rk, rv, inner, sk, sv, outer, out=(jk, jo, ji, cap, block)) == want
        rows = sort_rows(jk.download()[:want[0]], jo.download()[:want[0]], ji.download()[:want[0]])
        for a, b in zip(rows, materialised_rows(ik, iv, ok, ov)):
            assert np.array_equal(a, b)
        for c in (jk, jo, ji):
            c.free()
    for c in (rk, rv, sk_all, sv_all):
        c.free()


@pytest.mark.parametrize("algorithm", ["phj", "cpra"])
def test_oversize_build_partition_materialised(hj, algorithm):
    """A build key with 30 000 copies (seven table fills, dealt to several work items as fill groups) probed by
    200 copies: the 6 M rows of that key and all others are materialised and compared row by row."""
    rng = np.random.default_rng(4321)
    u = np.unique(rng.integers(1, 2**32, size=60_000, dtype=np.uint64).astype(np.uint32))
    hot = u[77]
    ik = np.concatenate([u, np.full(30_000, hot, np.uint32)])
    rng.shuffle(ik)
    iv = rng.integers(0, 2**32, size=len(ik), dtype=np.uint64).astype(np.uint32)
    ok = np.concatenate([u[rng.integers(0, len(u), size=150_000)], np.full(200, hot, np.uint32)])
    rng.shuffle(ok)
    ov = rng.integers(0, 2**32, size=len(ok), dtype=np.uint64).astype(np.uint32)
    want = numpy_join(ik, iv, ok, ov)
    assert want[0] > 6_000_000
    rk, rv, sk, sv = (hj.column(c) for c in (ik, iv, ok, ov))
    block = 4096
    cap = (want[0] // block + hj.device_info()["compute_units"] * 16 + 8) * block
    jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
    fn = hj.phj if algorithm == "phj" else hj.cpra
    prm = None if algorithm == "phj" else H.PhjParams(chunks=3)
    assert fn(rk, rv, len(ik), sk, sv, len(ok), prm, out=(jk, jo, ji, cap, block)) == want
    rows = sort_rows(jk.download()[:want[0]], jo.download()[:want[0]], ji.download()[:want[0]])
    for a, b in zip(rows, materialised_rows(ik, iv, ok, ov)):
        assert np.array_equal(a, b)
    for c in (rk, rv, sk, sv, jk, jo, ji):
        c.free()


def test_prepared_build_probed_by_a_batch_with_more_ranges_than_max_outer(hj):
    """The tiles per pass-1 range jump at 512-tile multiples, so a 12 M-row batch has MORE ranges (733) than the
    18 M rows the build side was prepared for (550): the per-range tables are sized for every batch up to
    max_outer (round 1 sized them for max_outer itself and the smaller batch wrote past them)."""
    inner, outer = 3_000_000, 30_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(21, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, fo, fi)
    want = (outer, sums[0], sums[1], sums[2])
    for prm in (None, H.PhjParams(fanout1=128, fanout2=8)):
        hj.phj_build(ik, iv, inner, 18_000_000, prm)
        total = (0, 0, 0, 0)
        for b, e in ((0, 12_000_000), (12_000_000, 30_000_000)):
            part = hj.phj_probe(ok.ptr + 4 * b, ov.ptr + 4 * b, e - b)
            assert part[0] == e - b
            total = tuple((x + y) & ((1 << 64) - 1) for x, y in zip(total, part))
        assert total == want
    for c in (ik, iv, ok, ov):
        c.free()


def test_two_contexts_on_two_host_threads_with_different_options(hj, oracle):
    """hjgpu.h: "thread-safe per context".  Two contexts, two host threads, overlapping joins, each context with
    its own options (one forces the chained tables and _UNIQUE): nothing on a launch path reads the environment
    or mutable process-wide state any more (round 1: a function-local static rewritten per call and eight
    getenv()s), so the results of one thread do not depend on what the other one does."""
    import threading
    ik, iv, ok, ov = oracle.generate(400_000, 1_600_000, seed=6)                 # 4 copies per build key
    want_all = numpy_join(ik, iv, ok, ov)
    hit = np.isin(ok, ik)
    want_unique = oracle.join_definition_unique(ik, iv, ok, ov)
    errors = []

    def worker(unique):
        try:
            with H.HjGpu() as ctx:
                if unique:
                    ctx.set_option("unique", 1)
                    ctx.set_option("force_chained", 1)
                    ctx.set_option("dense2", 1)
                rk, rv, sk, sv = (ctx.column(c) for c in (ik, iv, ok, ov))
                for i in range(12):
                    prm = H.PhjParams(fanout1=8 + i, fanout2=1 + i % 3, chunks=1 + i % 4)
                    for fn in (ctx.phj, ctx.cpra):
                        got = fn(rk, rv, len(ik), sk, sv, len(ok), prm)
                        if (got[:3] != want_unique) if unique else (got != want_all):
                            errors.append((unique, i, got))
                    got = ctx.npj(rk, rv, len(ik), sk, sv, len(ok))
                    if (got[:3] != want_unique) if unique else (got != want_all):
                        errors.append((unique, i, "npj", got))
                for c in (rk, rv, sk, sv):
                    c.free()
        except Exception as ex:          # surfaces in the main thread
            errors.append((unique, repr(ex)))

    threads = [threading.Thread(target=worker, args=(u,)) for u in (False, True)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    assert int(hit.sum()) == want_unique[0]


@pytest.mark.parametrize("selectivity,zipf", [(1.0, 0.0), (0.5, 0.0), (0.25, 1.2), (0.0, 0.0)])
def test_device_generator_selectivity_and_its_analytic_aggregates(hj, selectivity, zipf):
    """hjgpu_generate_select (write.cpp:1685-1689): join_d = d * selectivity common keys.  Small enough for numpy:
    unique non-zero build keys, the share of distinct probe keys with a partner is the selectivity, the aggregates
    accumulated during generation equal the join of the generated relations, and shards generated independently
    (the per-GPU ranges of a multi-GPU run) are the pieces of the whole."""
    inner, outer = 50_000, 400_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    want = hj.generate_select(7, inner, outer, 0, inner, 0, outer, fi, fo, zipf, selectivity, ik, iv, ok, ov)
    hik, hiv, hok, hov = ik.download(), iv.download(), ok.download(), ov.download()
    assert len(np.unique(hik)) == inner and (hik != 0).all() and (hok != 0).all()
    assert np.array_equal(hiv, hik * np.uint32(fi)) and np.array_equal(hov, hok * np.uint32(fo))
    assert want == numpy_join(hik, hiv, hok, hov)
    distinct_probe = np.unique(hok)
    assert len(distinct_probe) <= inner
    assert int(np.isin(distinct_probe, hik).sum()) == min(int(inner * selectivity), len(distinct_probe)) or zipf > 0
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    assert hj.npj(ik, iv, inner, ok, ov, outer) == want
    # three shards of the probe side, generated on their own
    total = (0, 0, 0, 0)
    for b, e in ((0, 100_000), (100_000, 100_016), (100_016, outer)):
        part = hj.generate_select(7, inner, outer, 0, 0, b, e - b, fi, fo, zipf, selectivity, None, None,
                                  ok.ptr + 4 * b, ov.ptr + 4 * b)
        total = tuple((x + y) & ((1 << 64) - 1) for x, y in zip(total, part))
    assert total == want and np.array_equal(ok.download(), hok)
    for c in (ik, iv, ok, ov):
        c.free()


@pytest.mark.parametrize("selectivity", [0.5, 0.0])
def test_full_size_64m_1g_with_non_matching_probe_keys(hj, selectivity):
    """BASELINE's size with write.cpp's selectivity below 1: half (none) of the probe side's distinct keys have a
    partner, so the probe kernels see misses at full size; the result must equal the aggregates accumulated while
    the relations were generated, for PHJ, CPRA and NPJ."""
    inner, outer = 64_000_000, 1_000_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    want = hj.generate_select(3, inner, outer, 0, inner, 0, outer, fi, fo, 0.0, selectivity, ik, iv, ok, ov)
    if selectivity == 0.0:
        assert want == (0, 0, 0, 0)
    else:
        assert 0.45 * outer < want[0] < 0.55 * outer
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    assert hj.cpra(ik, iv, inner, ok, ov, outer, H.PhjParams(chunks=8)) == want
    assert hj.npj(ik, iv, inner, ok, ov, outer) == want
    for c in (ik, iv, ok, ov):
        c.free()


@pytest.mark.parametrize("batch_tuples", [40_000, 300_000, 2_000_000])
def test_batched_probe_side_partitioning(hj, batch_tuples):
    """Two-pass plans partition the probe side in batches (pass 1 of a batch into a small reused buffer, pass 2
    straight from it: the intermediate copy stays in the Infinity Cache).  Forced here at test sizes through
    option "batch_tuples": ragged last batch, batches of a single range, unaligned probe columns, misses, duplicate
    build keys, materialised rows, prepared build side probed in pieces."""
    rng = np.random.default_rng(batch_tuples)
    base = np.unique(rng.integers(0, 2**32, size=400_000, dtype=np.uint64).astype(np.uint32))
    ik = np.concatenate([base, base[:30_000]])
    iv = rng.integers(0, 2**32, size=len(ik), dtype=np.uint64).astype(np.uint32)
    outer = 7_300_003
    ok = np.where(rng.random(outer) < 0.8, base[rng.integers(0, len(base), size=outer)],
                  rng.integers(0, 2**32, size=outer, dtype=np.uint64).astype(np.uint32)).astype(np.uint32)
    ov = rng.integers(0, 2**32, size=outer, dtype=np.uint64).astype(np.uint32)
    want = numpy_join(ik, iv, ok, ov)
    rk, rv = hj.column(ik), hj.column(iv)
    pad = 3                                             # probe columns start 12 bytes into their allocations
    sk_all = hj.column(np.concatenate([np.zeros(pad + 1, np.uint32), ok]))
    sv_all = hj.column(np.concatenate([np.zeros(pad + 1, np.uint32), ov]))
    hj.set_option("batch_tuples", batch_tuples)
    try:
        for off in (pad + 1, 0):                        # 16-byte aligned start, then the whole allocation (4 extra zeros join nothing... key 0 absent)
            sk, sv, n = sk_all.ptr + 4 * off, sv_all.ptr + 4 * off, outer + (pad + 1 - off)
            extra = (pad + 1 - off)
            w = want if extra == 0 else numpy_join(ik, iv, np.concatenate([np.zeros(extra, np.uint32), ok]), np.concatenate([np.zeros(extra, np.uint32), ov]))
            for prm in (H.PhjParams(fanout1=16, fanout2=9), H.PhjParams(fanout1=128, fanout2=3), None):
                assert hj.phj(rk, rv, len(ik), sk, sv, n, prm) == w
                st = hj.stats()
                if prm is not None:
                    assert st["batches"] >= 2 and st["fanout2"] > 1
        sk, sv = sk_all.ptr + 4 * (pad + 1), sv_all.ptr + 4 * (pad + 1)
        prm = H.PhjParams(fanout1=32, fanout2=5)
        block = 1024
        cap = (want[0] // block + hj.device_info()["compute_units"] * 16 + 8) * block
        jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
        assert hj.phj(rk, rv, len(ik), sk, sv, outer, prm, out=(jk, jo, ji, cap, block)) == want
        rows = sort_rows(jk.download()[:want[0]], jo.download()[:want[0]], ji.download()[:want[0]])
        for a, b in zip(rows, materialised_rows(ik, iv, ok, ov)):
            assert np.array_equal(a, b)
        # prepared build side, probe side in two pieces (each batched on its own)
        cut = 3_000_000 & ~15
        hj.phj_build(rk, rv, len(ik), outer - cut, prm)
        a = hj.phj_probe(sk, sv, cut)
        b = hj.phj_probe(sk + 4 * cut, sv + 4 * cut, outer - cut)
        assert tuple((x + y) & ((1 << 64) - 1) for x, y in zip(a, b)) == want
        for c in (jk, jo, ji):
            c.free()
    finally:
        hj.set_option("batch_tuples", 0)
    for c in (rk, rv, sk_all, sv_all):
        c.free()


def test_options_and_communicator_argument_errors(hj):
    """hjgpu_set_option / hjgpu_comm_*: unknown names, malformed values and impossible worlds are status codes
    (the reference asserts), and a refused option leaves the context as it was."""
    import hash_join_codes_knl_amd as H
    for name, value in (("no_such_option", "1"), ("unique", "yes"), ("join_cfg", "123,4,5"), ("join_cfg", "x"),
                        ("placement", "0"), ("placement", "99"), ("range_tiles", "-3"), ("scatter_cfg", "0,4"),
                        ("reserve_cus", "-1"), ("reserve_cus", "500")):
        with pytest.raises(H.HjGpuError) as e:
            hj.set_option(name, value)
        assert e.value.status == H.api.EINVAL
    hj.set_option("join_cfg", "512,13,2")
    ik = np.arange(1, 50_001, dtype=np.uint32)
    rk, rv = hj.column(ik), hj.column(ik)
    assert hj.phj(rk, rv, len(ik), rk, rv, len(ik)) == numpy_join(ik, ik, ik, ik)
    # K6 grids that leave CUs free (what the multi-GPU path sets so that RCCL's kernels find room): work is claimed
    # from ticket counters, so any grid finishes the passes - down to a single workgroup
    cus = hj.device_info()["compute_units"]
    for reserve in (16, min(128, cus - 1)):
        hj.set_option("reserve_cus", reserve)
        assert hj.phj(rk, rv, len(ik), rk, rv, len(ik), H.PhjParams(fanout1=24, fanout2=8)) == numpy_join(ik, ik, ik, ik)
        assert hj.cpra(rk, rv, len(ik), rk, rv, len(ik), H.PhjParams(chunks=3)) == numpy_join(ik, ik, ik, ik)
    hj.set_option("reserve_cus", 0)
    with pytest.raises(H.HjGpuError):
        H.HjComm.local(2, [0, 99], H.TRANSPORT_LOOPBACK)          # no such device
    with pytest.raises(H.HjGpuError):
        H.HjComm.local(2, [0, 0], H.TRANSPORT_RCCL)               # RCCL: one rank per device
    with H.HjComm.local(2, [0, 0], H.TRANSPORT_LOOPBACK) as comm:
        shards = [(rk, rv, len(ik), rk, rv, len(ik)), (None, None, len(ik) + 1, rk, rv, len(ik))]
        with pytest.raises(H.HjGpuError) as e:
            comm.phj_multi(shards, 0)                                  # |R| differs between the ranks
        assert e.value.status == H.api.EINVAL
        with pytest.raises(H.HjGpuError):
            comm.phj_multi(shards[:1] + [(None, None, len(ik), rk, rv, len(ik))], 5)    # root outside the world
        with pytest.raises(H.HjGpuError):
            comm.set_option("no_such_option", 1)
        with pytest.raises(H.HjGpuError):
            comm.set_option("reserve_cus", 129)
        comm.set_option("reserve_cus", 4)                          # applied to every rank's contexts
        shards = [(rk, rv, len(ik), rk, rv, len(ik)), (None, None, len(ik), rk, rv, len(ik))]
        want = numpy_join(ik, ik, np.concatenate([ik, ik]), np.concatenate([ik, ik]))
        assert comm.phj_multi(shards, 0)[0] == want
    for c in (rk, rv):
        c.free()


@pytest.mark.parametrize("chunks", [9, 16, 33, 64, 129, 256])
def test_cpra_with_more_chunks_than_eight(hj, chunks):
    """CPRA's chunks are the reference's #threads (cpra2.cpp:1757-1827, 2023: any thread count; its runs used 129 and more).
    Beyond 8 chunks the plan is always two passes with line-aligned final partitions (one region per partition whatever the
    number of chunks): build sides small enough for one pass, ragged sizes, an empty chunk tail, rows, _UNIQUE."""
    rng = np.random.default_rng(1000 + chunks)
    for inner, outer in ((3000, 70_001), (250_000, 1_000_003), (chunks * 16 - 5, 5000)):
        u = np.unique(rng.integers(1, 2**32, size=inner, dtype=np.uint64).astype(np.uint32))
        ik = rng.permutation(u)
        iv = rng.integers(0, 2**32, size=len(ik), dtype=np.uint64).astype(np.uint32)
        ok = np.where(rng.random(outer) < 0.8, ik[rng.integers(0, len(ik), size=outer)], rng.integers(1, 2**32, size=outer, dtype=np.uint64).astype(np.uint32)).astype(np.uint32)
        ov = rng.integers(0, 2**32, size=outer, dtype=np.uint64).astype(np.uint32)
        want = numpy_join(ik, iv, ok, ov)
        rk, rv, sk, sv = (hj.column(c) for c in (ik, iv, ok, ov))
        assert hj.cpra(rk, rv, len(ik), sk, sv, outer, H.PhjParams(chunks=chunks)) == want
        assert hj.cpra(rk, rv, len(ik), sk, sv, outer, H.PhjParams(chunks=chunks, flags=H.FLAG_UNIQUE)) == want
        assert hj.cpra(rk, rv, len(ik), sk, sv, outer, H.PhjParams(chunks=chunks, fanout1=24, fanout2=7)) == want
        for c in (rk, rv, sk, sv):
            c.free()
    # an explicit single-pass plan cannot hold more than 8 chunks; neither can the dense final layout
    ik = np.arange(1, 2001, dtype=np.uint32)
    rk = hj.column(ik)
    with pytest.raises(H.HjGpuError):
        hj.cpra(rk, rk, len(ik), rk, rk, len(ik), H.PhjParams(chunks=chunks, fanout1=64, fanout2=1))
    with pytest.raises(H.HjGpuError):
        hj.cpra(rk, rk, len(ik), rk, rk, len(ik), H.PhjParams(chunks=257))
    rk.free()


def test_full_size_cpra_in_64_chunks(hj):
    """64 M x 1 G in 64 and in 129 chunks (./cpra 129 ...: the reference's own thread count, cpra2.cpp:2023): the same aggregates as in 8 chunks and as PHJ"""
    inner, outer = 64_000_000, 1_000_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(5, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, fo, fi)
    want = (outer, sums[0], sums[1], sums[2])
    for chunks in (8, 64, 37, 129):
        assert hj.cpra(ik, iv, inner, ok, ov, outer, H.PhjParams(chunks=chunks)) == want
    for c in (ik, iv, ok, ov):
        c.free()


def test_option_solo_changes_stores_not_results(oracle):
    """Option solo (the caller's promise that nothing else runs on the device beside the context's blocking joins): K6's
    partial-line stores and the result rows stay plain instead of non-temporal - same aggregates, same rows, all three
    algorithms; the enqueue-only forms ignore it (they always write non-temporal)."""
    ik, iv, ok, ov = oracle.generate(600_000, 150_000, seed=77)
    want = numpy_join(ik, iv, ok, ov)
    with H.HjGpu() as ctx:
        rk, rv, sk, sv = (ctx.column(c) for c in (ik, iv, ok, ov))
        d = ctx.column(4, np.uint64)
        for solo in ("0", "1"):
            ctx.set_option("solo", solo)
            for algo, prm in (("phj", None), ("cpra", H.PhjParams(chunks=5)), ("npj", None)):
                assert getattr(ctx, algo)(rk, rv, len(ik), sk, sv, len(ok), prm) == want
                getattr(ctx, algo + "_async")(rk, rv, len(ik), sk, sv, len(ok), prm, d)
                ctx.synchronize()
                assert tuple(int(x) for x in d.download()) == want
            block = 1024
            cap = (want[0] // block + ctx.device_info()["compute_units"] * 16 + 8) * block
            jk, jo, ji = ctx.column(cap), ctx.column(cap), ctx.column(cap)
            assert ctx.phj(rk, rv, len(ik), sk, sv, len(ok), out=(jk, jo, ji, cap, block)) == want
            rows = sort_rows(jk.download()[:want[0]], jo.download()[:want[0]], ji.download()[:want[0]])
            for a, b in zip(rows, materialised_rows(ik, iv, ok, ov)):
                assert np.array_equal(a, b)
            for c in (jk, jo, ji):
                c.free()
        for c in (rk, rv, sk, sv, d):
            c.free()
