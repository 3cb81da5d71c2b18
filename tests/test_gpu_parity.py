"""GPU parity tests: every HIP operator / whole join, called through the C-ABI
(include/hjgpu.h via ctypes), against the CPU oracle on the same seeded inputs.
Integer work: the bar is bit-exact."""
import os

import numpy as np
import pytest

import hash_join_codes_knl_amd as H
from helpers import mulhi_hash, pairs, numpy_join, materialised_rows, sort_rows

pytestmark = pytest.mark.gpu

F_A, F_B = 0x9E3779B1, 0x85EBCA6B


def _cols(hj, *arrays):
    return [hj.column(a) for a in arrays]


def _free(*cols):
    for c in cols:
        c.free()


# ---------------------------------------------------------------- K4 histogram
@pytest.mark.parametrize("n,fanout", [(0, 8), (1, 1), (5, 3), (1000, 7), (100003, 64),
                                      (1 << 20, 1000), (3_000_017, 17000), (50_000, 32768)])
def test_histogram_matches_oracle(hj, oracle, n, fanout):
    rng = np.random.default_rng(n + fanout)
    keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    want = oracle.histogram(keys, F_A, fanout).astype(np.uint64)
    dk = hj.column(keys) if n else hj.column(1)
    dc = hj.column(fanout, np.uint64)
    hj.histogram(dk, n, F_A, fanout, dc)
    got = dc.download()
    _free(dk, dc)
    assert got.sum() == n
    assert np.array_equal(got, want)


def test_histogram_unaligned_view(hj, oracle):
    """Sub-ranges that do not start on a 16-byte boundary (pass-2 segments do not)."""
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 2**32, size=10_000, dtype=np.uint64).astype(np.uint32)
    dk = hj.column(keys)
    dc = hj.column(37, np.uint64)
    for off, n in [(1, 999), (2, 4097), (3, 1), (7, 8190), (5, 0)]:
        hj.histogram(dk.ptr + 4 * off, n, F_B, 37, dc)
        assert np.array_equal(dc.download(), oracle.histogram(keys[off:off + n], F_B, 37).astype(np.uint64))
    _free(dk, dc)


# ---------------------------------------------------------------- K5+K6 partition
@pytest.mark.parametrize("n,fanout", [(0, 4), (1, 2), (17, 5), (8192, 64), (8193, 64),
                                      (100_003, 128), (1_000_000, 1000), (2_500_000, 1024),
                                      (300_000, 1)])
def test_partition_matches_oracle(hj, oracle, n, fanout):
    rng = np.random.default_rng(n * 31 + fanout)
    keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    vals = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    counts, ko_ref, vo_ref = oracle.partition(keys, vals, F_A, fanout)
    dk, dv = (hj.column(keys), hj.column(vals)) if n else (hj.column(1), hj.column(1))
    dko, dvo = hj.column(max(n, 1)), hj.column(max(n, 1))
    doff = hj.column(fanout + 1, np.uint64)
    hj.partition(dk, dv, n, F_A, fanout, dko, dvo, doff)
    off = doff.download().astype(np.int64)
    ko, vo = dko.download(n), dvo.download(n)
    _free(dk, dv, dko, dvo, doff)
    want_off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
    assert np.array_equal(off, want_off)                      # histogram + prefix: exact
    # every tuple lies in its hash partition
    part_of_pos = np.searchsorted(off, np.arange(n), side="right") - 1
    assert np.array_equal(mulhi_hash(ko, F_A, fanout), part_of_pos)
    # per-partition multiset equality with the oracle (order inside a partition is free)
    got = pairs(ko, vo); ref = pairs(ko_ref, vo_ref)
    key = part_of_pos.astype(np.uint64)
    got_sorted = got[np.lexsort((got, key))]
    ref_sorted = ref[np.lexsort((ref, key))]
    assert np.array_equal(got_sorted, ref_sorted)


def test_partition_skewed_all_one_partition(hj, oracle):
    n = 50_000
    keys = np.full(n, 12345, np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    dk, dv, dko, dvo = hj.column(keys), hj.column(vals), hj.column(n), hj.column(n)
    doff = hj.column(65, np.uint64)
    hj.partition(dk, dv, n, F_A, 64, dko, dvo, doff)
    off = doff.download()
    p = int(mulhi_hash(keys[:1], F_A, 64)[0])
    assert off[p] == 0 and off[p + 1] == n
    assert np.array_equal(np.sort(dvo.download()), vals)
    assert (dko.download() == 12345).all()
    _free(dk, dv, dko, dvo, doff)


# ---------------------------------------------------------------- whole joins
CASES = {
    # name: (outer, inner, selectivity, seed)
    "tiny": (10, 3, 1.0, 1),
    "small_unique": (4096, 1024, 1.0, 2),
    "mid_unique": (65_536, 4_096, 1.0, 3),
    "build_dups_16x": (1_000, 16_000, 1.0, 4),         # config-1 shape: 16 copies per build key
    "selectivity_half": (50_000, 20_000, 0.5, 5),
    "selectivity_zero": (5_000, 5_000, 0.0, 6),
    "large": (2_000_000, 250_000, 1.0, 7),
}


def _join_all(hj, ik, iv, ok, ov, phj_params=None, npj_params=None, algos=("npj", "phj", "cpra")):
    rk, rv, sk, sv = _cols(hj, ik if len(ik) else np.zeros(1, np.uint32), iv if len(iv) else np.zeros(1, np.uint32),
                           ok if len(ok) else np.zeros(1, np.uint32), ov if len(ov) else np.zeros(1, np.uint32))
    out = {}
    if "npj" in algos:
        out["npj"] = hj.npj(rk, rv, len(ik), sk, sv, len(ok), npj_params)
    if "phj" in algos:
        out["phj"] = hj.phj(rk, rv, len(ik), sk, sv, len(ok), phj_params)
    if "cpra" in algos:
        out["cpra"] = hj.cpra(rk, rv, len(ik), sk, sv, len(ok), phj_params)
    _free(rk, rv, sk, sv)
    return out


@pytest.mark.parametrize("name", list(CASES))
def test_joins_match_oracle(hj, oracle, name):
    outer, inner, sel, seed = CASES[name]
    ik, iv, ok, ov = oracle.generate(outer, inner, selectivity=sel, seed=seed)
    want = oracle.join_definition(ik, iv, ok, ov)
    assert want == numpy_join(ik, iv, ok, ov)
    # the oracle's own three algorithms agree with the definition (restated reference)
    assert oracle.npj(ik, iv, ok, ov, threads=2) == want
    assert oracle.phj(ik, iv, ok, ov, threads=2, hash_table_limit=max(8, inner // 50)) == want
    got = _join_all(hj, ik, iv, ok, ov)
    for algo, res in got.items():
        assert res == want, "%s on %s: got %r want %r" % (algo, name, res, want)


@pytest.mark.parametrize("f1,f2", [(2, 1), (3, 1), (64, 1), (5, 7), (64, 64), (128, 100), (1000, 32)])
def test_phj_cpra_any_fanout(hj, oracle, f1, f2):
    ik, iv, ok, ov = oracle.generate(200_000, 30_000, seed=11)
    want = oracle.join_definition(ik, iv, ok, ov)
    for chunks in (1, 3, 8):
        prm = H.PhjParams(fanout1=f1, fanout2=f2, chunks=chunks)
        got = _join_all(hj, ik, iv, ok, ov, phj_params=prm, algos=("phj", "cpra"))
        assert got["phj"] == want and got["cpra"] == want, (f1, f2, chunks, got, want)


def test_phj_overflow_partitions_multi_fill(hj, oracle):
    """Partitions far larger than one LDS table fill (duplicate-heavy build side):
    the join must re-fill the table and re-stream the probe slice."""
    ik, iv, ok, ov = oracle.generate(3_000, 96_000, seed=12)       # 32 copies per key
    want = oracle.join_definition(ik, iv, ok, ov)
    prm = H.PhjParams(fanout1=2, fanout2=1)                        # ~48K build rows per partition
    got = _join_all(hj, ik, iv, ok, ov, phj_params=prm, algos=("phj", "cpra"))
    assert got["phj"] == want and got["cpra"] == want


def test_phj_key_zero_is_legal(hj, oracle):
    """PHJ/CPRA accept key 0 (per-partition sentinel, phj.cpp:1886-1897)."""
    ik, iv, ok, ov = oracle.generate(20_000, 5_000, seed=13)
    ik = ik.copy(); ok = ok.copy()
    victim = ik[0]
    ik[0] = 0; iv[0] = 777
    ok[ok == victim] = 0
    ok[:5] = 0
    ov = (ok.astype(np.uint64) * 3 + 1).astype(np.uint32)
    want = numpy_join(ik, iv, ok, ov)
    assert want[0] > 0
    got = _join_all(hj, ik, iv, ok, ov, algos=("phj", "cpra"))
    assert got["phj"] == want and got["cpra"] == want


def test_npj_rejects_key_zero(hj, oracle):
    ik, iv, ok, ov = oracle.generate(1000, 1000, seed=14)
    ik = ik.copy(); ik[10] = 0
    with pytest.raises(H.HjGpuError) as e:
        _join_all(hj, ik, iv, ok, ov, algos=("npj",))
    assert e.value.status == H.api.EZEROKEY


def test_empty_sides(hj, oracle):
    ik, iv, ok, ov = oracle.generate(1000, 500, seed=15)
    e = np.zeros(0, np.uint32)
    for a, b, c, d in [(e, e, ok, ov), (ik, iv, e, e), (e, e, e, e)]:
        got = _join_all(hj, a, b, c, d)
        for res in got.values():
            assert res == (0, 0, 0, 0)


@pytest.mark.parametrize("load", [0.25, 0.5, 0.9])
def test_npj_load_factor_is_free(hj, oracle, load):
    ik, iv, ok, ov = oracle.generate(100_000, 40_000, seed=16)
    want = oracle.npj(ik, iv, ok, ov, threads=2, load=0.9)        # reference load (npj.cpp:944)
    got = _join_all(hj, ik, iv, ok, ov, npj_params=H.NpjParams(load=load), algos=("npj",))
    assert got["npj"] == want


def test_npj_table_is_reference_format(hj, oracle):
    """A table built on the GPU is probe-able by the oracle's probe and vice versa:
    same bucket format (val<<32|key), same hash, same walk (npj.cpp:190-212, 412-445)."""
    import ctypes as C
    ik, iv, ok, ov = oracle.generate(30_000, 10_000, seed=17)
    buckets = int(len(ik) / 0.5)
    want = oracle.join_definition(ik, iv, ok, ov)
    rk, rv, sk, sv = _cols(hj, ik, iv, ok, ov)
    dt = hj.column(buckets, np.uint64)
    hj.npj_build(rk, rv, len(ik), dt, buckets, F_A)
    table = dt.download()
    # GPU table -> oracle probe
    r = oracle.Result()
    oracle.lib().hjo_npj_probe(ok, ov, len(ok), table, buckets, F_A, 0, C.byref(r), None, None)
    assert r.as_tuple() == want
    # same multiset of buckets as the oracle's build (placement may differ by insert order)
    t2 = np.zeros(buckets, np.uint64)
    oracle.lib().hjo_npj_build(ik, iv, len(ik), t2, buckets, F_A, 0)
    assert np.array_equal(np.sort(table), np.sort(t2))
    # oracle table -> GPU probe
    dt.upload(t2)
    assert hj.npj_probe(sk, sv, len(ok), dt, buckets, F_A) == want
    _free(rk, rv, sk, sv, dt)


# ---------------------------------------------------------------- materialised output (K9)
@pytest.mark.parametrize("algo", ["npj", "phj", "cpra"])
@pytest.mark.parametrize("case", ["mid_unique", "build_dups_16x", "selectivity_half"])
def test_materialised_rows_match(hj, oracle, algo, case):
    outer, inner, sel, seed = CASES[case]
    ik, iv, ok, ov = oracle.generate(outer, inner, selectivity=sel, seed=seed)
    want = oracle.join_definition(ik, iv, ok, ov)
    wk, wo, wi = materialised_rows(ik, iv, ok, ov)
    block = 256
    capacity = (want[0] // block + 8192 + 8) * block
    rk, rv, sk, sv = _cols(hj, ik, iv, ok, ov)
    jk, jo, ji = hj.column(capacity), hj.column(capacity), hj.column(capacity)
    res = getattr(hj, algo)(rk, rv, len(ik), sk, sv, len(ok), None, out=(jk, jo, ji, capacity, block))
    assert res == want
    n = res[0]
    gk, go, gi = sort_rows(jk.download(n), jo.download(n), ji.download(n))
    _free(rk, rv, sk, sv, jk, jo, ji)
    assert np.array_equal(gk, wk) and np.array_equal(go, wo) and np.array_equal(gi, wi)


def test_materialised_overflow_is_reported(hj, oracle):
    ik, iv, ok, ov = oracle.generate(50_000, 10_000, seed=21)
    rk, rv, sk, sv = _cols(hj, ik, iv, ok, ov)
    cap = 256 * 4
    jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
    with pytest.raises(H.HjGpuError) as e:
        hj.phj(rk, rv, len(ik), sk, sv, len(ok), None, out=(jk, jo, ji, cap, 256))
    assert e.value.status == H.api.EOVERFLOW
    _free(rk, rv, sk, sv, jk, jo, ji)


# ---------------------------------------------------------------- argument checking
def test_bad_arguments(hj):
    d = hj.column(np.arange(64, dtype=np.uint32))
    with pytest.raises(H.HjGpuError) as e:
        hj.phj(d.ptr + 4, d.ptr + 4, 8, d, d, 8)                   # misaligned column
    assert e.value.status == H.api.EALIGN
    with pytest.raises(H.HjGpuError) as e:
        hj.phj(d, d, 8, d, d, 8, H.PhjParams(fanout1=4, fanout2=1, factor1=2))   # even factor
    assert e.value.status == H.api.EINVAL
    with pytest.raises(H.HjGpuError) as e:
        hj.phj(d, d, 8, d, d, 8, H.PhjParams(fanout1=2000, fanout2=1))
    assert e.value.status == H.api.EINVAL
    d.free()


# ---------------------------------------------------------------- generator
def test_device_generator_contract(hj):
    inner, outer = 100_000, 1_000_000
    fi, fo = 0x0F0F0F0F | 1, 0x12345679
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(42, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    hik, hiv, hok, hov = ik.download(), iv.download(), ok.download(), ov.download()
    assert (hik != 0).all() and len(np.unique(hik)) == inner          # unique non-zero build keys
    assert np.array_equal(hiv, (hik.astype(np.uint64) * fi).astype(np.uint32))
    assert np.array_equal(hov, (hok.astype(np.uint64) * fo).astype(np.uint32))
    assert np.isin(hok, hik).all()                                     # selectivity 1
    assert len(np.unique(hok)) == inner                                # every build key probed
    # shards are position-pure: generating [a, a+c) alone gives the same tuples
    sk, sv = hj.column(1000), hj.column(1000)
    hj.generate(42, inner, outer, 123_456, 1000, fi, fo, None, None, sk, sv)
    assert np.array_equal(sk.download(), hok[123_456:124_456])
    # analytic aggregates == join result
    want = numpy_join(hik, hiv, hok, hov)
    sums = hj.column_sums(ok, outer, fo, fi)
    assert (outer, sums[0], sums[1], sums[2]) == want
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    _free(ik, iv, ok, ov, sk, sv)


def test_device_generator_duplicate_build_side(hj):
    """outer < inner: min(inner, outer) distinct keys, build side repeats them (write.cpp:1687-1689)."""
    inner, outer = 160_000, 10_000
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(7, inner, outer, 0, outer, 3, 5, ik, iv, ok, ov)
    hik, hiv, hok, hov = ik.download(), iv.download(), ok.download(), ov.download()
    assert len(np.unique(hik)) == outer and len(np.unique(hok)) == outer
    want = numpy_join(hik, hiv, hok, hov)
    assert want[0] == inner
    assert hj.npj(ik, iv, inner, ok, ov, outer) == want
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    assert hj.cpra(ik, iv, inner, ok, ov, outer) == want
    _free(ik, iv, ok, ov)


# ---------------------------------------------------------------- overlapped build-side arrival
def test_phj_overlapped_waits_for_build_side(hj, oracle):
    """hjgpu_phj_overlapped_async: R is produced on a side stream AFTER the join was
    enqueued; the library must not read it before the event fires."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    ik, iv, ok, ov = oracle.generate(1_000_000, 200_000, seed=31)
    want = oracle.join_definition(ik, iv, ok, ov)
    t = lambda a: torch.from_numpy(np.concatenate([a, np.zeros(4, np.uint32)]).view(np.int32)).to(dev)
    src_k, src_v, sk, sv = t(ik), t(iv), t(ok), t(ov)
    rk, rv = torch.zeros_like(src_k), torch.zeros_like(src_v)        # garbage until the side stream ran
    d_res = torch.zeros(4, dtype=torch.int64, device=dev)
    hj.reserve(len(ik), len(ok))
    main, side = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
    big = torch.empty(256 * 1024 * 1024, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(20):
            big.add_(1)                                              # ~ms of delay before R exists
        rk.copy_(src_k); rv.copy_(src_v)
        ready = torch.cuda.Event()
        ready.record(side)
    hj.phj_overlapped_async(rk.data_ptr(), rv.data_ptr(), len(ik), sk.data_ptr(), sv.data_ptr(), len(ok),
                            None, d_res.data_ptr(), main.cuda_stream, ready.cuda_event)
    torch.cuda.synchronize()
    got = tuple(int(x) & ((1 << 64) - 1) for x in d_res.tolist())
    assert got == want
    assert hj.stats()["ms_inner_wait"] > 0.0


# ---------------------------------------------------------------- operator-level K7+K8, host path, async forms
def test_join_partitions_on_oracle_partitioned_columns(hj, oracle):
    """hjgpu_join_partitions consumes co-partitioned columns + offsets produced elsewhere
    (here: by the CPU oracle's partition()) - the reference's operator-level seam
    (phj.cpp:1869-1924): partition(...) then build/probe per partition."""
    ik, iv, ok, ov = oracle.generate(300_000, 60_000, seed=41)
    want = oracle.join_definition(ik, iv, ok, ov)
    f, F = 0x9E3779B1, 97
    rc, rk, rv = oracle.partition(ik, iv, f, F)
    sc, sk, sv = oracle.partition(ok, ov, f, F)
    roff = np.concatenate([np.zeros(1, np.uint64), np.cumsum(rc, dtype=np.uint64)])
    soff = np.concatenate([np.zeros(1, np.uint64), np.cumsum(sc, dtype=np.uint64)])
    cols = [hj.column(x) for x in (rk, rv, sk, sv)]
    d_roff, d_soff = hj.column(roff, np.uint64), hj.column(soff, np.uint64)
    prm = H.PhjParams(fanout1=F, fanout2=1, factor1=f)
    assert hj.join_partitions(cols[0], cols[1], d_roff, cols[2], cols[3], d_soff, prm) == want
    # GPU-partitioned columns (hjgpu_partition x2) feed the same operator
    pk, pv, poff = hj.column(len(ik)), hj.column(len(ik)), hj.column(F + 1, np.uint64)
    qk, qv, qoff = hj.column(len(ok)), hj.column(len(ok)), hj.column(F + 1, np.uint64)
    raw = [hj.column(x) for x in (ik, iv, ok, ov)]
    hj.partition(raw[0], raw[1], len(ik), f, F, pk, pv, poff)
    hj.partition(raw[2], raw[3], len(ok), f, F, qk, qv, qoff)
    assert hj.join_partitions(pk, pv, poff, qk, qv, qoff, prm) == want
    # two-level layout (p1 * F2 + p2) built on the host
    f2, F1, F2 = 0x85EBCA6B, 13, 11
    def two_level(k, v):
        pid = mulhi_hash(k, f, F1) * F2 + mulhi_hash(k, f2, F2)
        order = np.argsort(pid, kind="stable")
        off = np.concatenate([np.zeros(1, np.uint64), np.cumsum(np.bincount(pid, minlength=F1 * F2), dtype=np.uint64)])
        return k[order], v[order], off
    rk2, rv2, roff2 = two_level(ik, iv)
    sk2, sv2, soff2 = two_level(ok, ov)
    c2 = [hj.column(x) for x in (rk2, rv2, sk2, sv2)]
    o2 = [hj.column(roff2, np.uint64), hj.column(soff2, np.uint64)]
    prm2 = H.PhjParams(fanout1=F1, fanout2=F2, factor1=f, factor2=f2)
    assert hj.join_partitions(c2[0], c2[1], o2[0], c2[2], c2[3], o2[1], prm2) == want
    _free(*cols, d_roff, d_soff, pk, pv, poff, qk, qv, qoff, *raw, *c2, *o2)


@pytest.mark.parametrize("algorithm", [0, 1, 2])
def test_join_host_path(hj, oracle, algorithm):
    """hjgpu_join_host: what the npj/phj/cpra mains call after their fread()s."""
    ik, iv, ok, ov = oracle.generate(150_000, 40_000, seed=42)
    want = oracle.join_definition(ik, iv, ok, ov)
    got, stats = hj.join_host(algorithm, ik, iv, ok, ov)
    assert got == want
    assert stats["ms_total"] > 0 and stats["ms_join"] > 0
    if algorithm:
        assert stats["fanout1"] * stats["fanout2"] >= 2
    else:
        assert stats["buckets"] > len(ik)


def test_async_entry_points_write_device_result(hj, oracle):
    ik, iv, ok, ov = oracle.generate(400_000, 90_000, seed=43)
    want = oracle.join_definition(ik, iv, ok, ov)
    rk, rv, sk, sv = _cols(hj, ik, iv, ok, ov)
    hj.reserve(len(ik), len(ok))
    for fn, prm in ((hj.phj_async, None), (hj.cpra_async, H.PhjParams(chunks=4)), (hj.npj_async, None)):
        d_res = hj.column(np.zeros(4, np.uint64), np.uint64)
        fn(rk, rv, len(ik), sk, sv, len(ok), prm, d_res)
        hj.synchronize()
        assert tuple(int(x) for x in d_res.download()) == want
        d_res.free()
    _free(rk, rv, sk, sv)


def test_generate_range_is_position_pure(hj):
    inner, outer = 50_000, 300_000
    full = [hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)]
    hj.generate(5, inner, outer, 0, outer, 7, 9, *full)
    hik, hiv = full[0].download(), full[1].download()
    part_k, part_v = hj.column(1000), hj.column(1000)
    hj.generate_range(5, inner, outer, 12_345, 1000, 0, 0, 7, 9, part_k, part_v, None, None)
    assert np.array_equal(part_k.download(), hik[12_345:13_345])
    assert np.array_equal(part_v.download(), hiv[12_345:13_345])
    _free(*full, part_k, part_v)


@pytest.mark.parametrize("n,skew", [(0, 0), (1, 0), (3, 1), (5, 3), (1027, 2), (1_000_003, 1), (4_000_000, 0)])
def test_column_sums_any_alignment(hj, n, skew):
    rng = np.random.default_rng(n + skew)
    host = rng.integers(0, 2**32, n + skew + 4, dtype=np.uint64).astype(np.uint32)
    col = hj.column(host)
    k = host[skew:skew + n].astype(np.uint64)
    fa, fb = 0x9E3779B1, 0x85EBCA6B
    want = (int(k.sum()), int(((k * fa) & 0xFFFFFFFF).sum()), int(((k * fb) & 0xFFFFFFFF).sum()))
    assert hj.column_sums(col.ptr + 4 * skew, n, fa, fb) == want
    col.free()


@pytest.mark.parametrize("algorithm", [0, 1, 2])
def test_join_host_from_pinned_columns(hj, oracle, algorithm):
    """hjgpu_host_alloc + hjgpu_join_host: page-locked columns are DMA'd directly (the host
    programs' fread targets); same result as from pageable memory, upload time reported.
    3 M probe tuples: more than one 32 MiB staging buffer's worth on the pageable path."""
    ik, iv, ok, ov = oracle.generate(9_000_000, 300_000, seed=44)
    want = oracle.join_definition(ik, iv, ok, ov)
    pinned = [hj.host_column(len(c)) for c in (ik, iv, ok, ov)]
    for dst, src in zip(pinned, (ik, iv, ok, ov)):
        dst.array[:] = src
    got_pinned, st = hj.join_host(algorithm, *pinned)
    got_pageable, st2 = hj.join_host(algorithm, ik, iv, ok, ov)
    for c in pinned:
        c.free()
    assert got_pinned == want and got_pageable == want
    assert st["ms_upload"] > 0 and st2["ms_upload"] > 0


@pytest.mark.parametrize("algorithm", [0, 1, 2])
@pytest.mark.parametrize("batch", [4096, 65_536, 1_000_000])
def test_join_host_in_batches_behind_the_upload(hj, oracle, algorithm, batch):
    """hjgpu_join_host, aggregates only: the build side is prepared once and the probe side travels in batches of
    option "host_batch" rows through two device buffers, batch i joined while batch i + 1 is uploaded (R join S = union
    over the batches; phj.cpp:1869-1924, cpra2.cpp:1757-1827, npj.cpp:882-901).  Ragged last batch, page-locked and
    pageable columns, duplicates on the build side, _UNIQUE, and NPJ's reserved key on the build side."""
    with H.HjGpu() as ctx:
        ctx.set_option("host_batch", batch)
        ik, iv, ok, ov = oracle.generate(2_500_037, 120_011, seed=batch % 89)      # 2.5 M probe rows, 120 K build rows
        want = oracle.join_definition(ik, iv, ok, ov)
        got, st = ctx.join_host(algorithm, ik, iv, ok, ov)
        assert got == want
        assert st["batches"] == -(-len(ok) // ((batch + 15) // 16 * 16)) and st["ms_upload"] > 0 and st["ms_join"] > 0
        pinned = [ctx.host_column(len(c)) for c in (ik, iv, ok, ov)]
        for dst, src in zip(pinned, (ik, iv, ok, ov)):
            dst.array[:] = src
        assert ctx.join_host(algorithm, *pinned)[0] == want
        for c in pinned:
            c.free()
        # build keys repeat (4 copies): every probe tuple matches four times; _UNIQUE: once
        ik2, iv2, ok2, ov2 = oracle.generate(300_000, 1_200_000, seed=7)
        assert ctx.join_host(algorithm, ik2, iv2, ok2, ov2)[0] == oracle.join_definition(ik2, iv2, ok2, ov2)
        U = H.api.FLAG_UNIQUE
        got_u = ctx.join_host(algorithm, ik2, iv2, ok2, ov2, H.PhjParams(flags=U), H.NpjParams(flags=U))[0]
        assert got_u[:3] == oracle.join_definition_unique(ik2, iv2, ok2, ov2)
        # fewer than two batches: the whole probe side at once, as before
        assert ctx.join_host(algorithm, ik, iv, ok[:batch], ov[:batch])[1]["batches"] == 0
        if algorithm == 0:
            bad = ik.copy()
            bad[len(bad) // 2] = 0
            with pytest.raises(H.HjGpuError) as e:
                ctx.join_host(0, bad, iv, ok, ov)
            assert e.value.status == H.api.EZEROKEY


def test_host_programs_end_to_end(hj, oracle, tmp_path):
    """./write -> ./npj ./phj ./cpra on the GPU box: same CLI, files and stdout formats as the
    reference's programs (npj.cpp:1114, phj.cpp:2197, cpra2.cpp:1984/2208); the aggregates on
    stderr equal the oracle's join of the generated files."""
    import subprocess
    lib = os.path.join(os.path.dirname(os.path.abspath(H.__file__)), "lib")
    env = dict(os.environ, HJ_SEED="11")
    subprocess.check_call([os.path.join(lib, "write"), "4", "400000", "90000"], cwd=tmp_path, env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cols = [np.fromfile(tmp_path / ("%s_%d.txt" % (p, n)), dtype="<u4")
            for p, n in (("ik", 90000), ("iv", 90000), ("ok", 400000), ("ov", 400000))]
    want = oracle.join_definition(*cols)
    for prog in ("npj", "phj", "cpra"):
        p = subprocess.run([os.path.join(lib, prog), "8", "400000", "90000"], cwd=tmp_path,
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert ("join_tuples=%d sum_keys=%d sum_outer_vals=%d sum_inner_vals=%d" % want) in p.stderr
        assert "upload of the four columns" in p.stderr
        lines = p.stdout.strip().splitlines()
        if prog == "npj":
            assert len(lines) == 1 and float(lines[0]) > 0
        elif prog == "phj":
            assert len(lines[0].split("\t")) >= 3
        else:
            assert lines[0].startswith("copy:\t") and float(lines[1]) > 0
    # ./cpra 129: the reference takes any #threads (cpra2.cpp:2023; its runs used 129 and more); up to 256 chunks are used as asked, more are capped with a note
    for threads, note in ((64, False), (129, False), (300, True)):
        p = subprocess.run([os.path.join(lib, "cpra"), str(threads), "400000", "90000"], cwd=tmp_path, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert ("join_tuples=%d sum_keys=%d sum_outer_vals=%d sum_inner_vals=%d" % want) in p.stderr
        assert ("chunks requested" in p.stderr) == note, p.stderr


def test_configs0_npj_64_1000000_16000000_on_written_files(hj, oracle, tmp_path):
    """BASELINE.json configs[0], the reference's own invocation (npj.cpp:929-947): `./write` generates the relations (1 M probe
    tuples, 16 M build tuples = 16 copies of every key), `./npj 64 1000000 16000000` joins the files on the GPU; count and the three
    checksums equal the oracle's restatement of run() (npj.cpp:769-927, load 0.90) on the same files and the join's definition."""
    import subprocess
    lib = os.path.join(os.path.dirname(os.path.abspath(H.__file__)), "lib")
    subprocess.check_call([os.path.join(lib, "write"), "1", "1000000", "16000000"], cwd=tmp_path, env=dict(os.environ, HJ_SEED="1"),
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ik, iv, ok, ov = [np.fromfile(tmp_path / ("%s_%d.txt" % (p, n)), dtype="<u4")
                      for p, n in (("ik", 16000000), ("iv", 16000000), ("ok", 1000000), ("ov", 1000000))]
    want = oracle.npj(ik, iv, ok, ov, threads=4, load=0.90)
    assert want == oracle.join_definition(ik, iv, ok, ov) and want[0] == 16_000_000
    p = subprocess.run([os.path.join(lib, "npj"), "64", "1000000", "16000000"], cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert ("join_tuples=%d sum_keys=%d sum_outer_vals=%d sum_inner_vals=%d" % want) in p.stderr, p.stderr
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1 and float(lines[0]) > 0              # the reference prints the seconds of the join (npj.cpp:1114)
    # ... and the same relations through the C-ABI on resident columns, rows materialised: 16 rows per probe tuple
    rk, rv, sk, sv = (hj.column(c) for c in (ik, iv, ok, ov))
    assert hj.npj(rk, rv, len(ik), sk, sv, len(ok), H.NpjParams(load=0.90)) == want
    assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == want and hj.cpra(rk, rv, len(ik), sk, sv, len(ok)) == want
    for c in (rk, rv, sk, sv):
        c.free()


def test_host_programs_on_several_ranks(hj, oracle, tmp_path):
    """The multi-GPU path of the C++ hosts (every visible GPU takes a share: hjgpu_comm_create_local +
    hjgpu_join_host_multi), driven here as 3 loopback ranks on the one GPU of the test box and - through RCCL
    from C++ - as HJGPU_DEVICES=0 plus a second rank that the box does not have (must fail cleanly)."""
    import subprocess
    lib = os.path.join(os.path.dirname(os.path.abspath(H.__file__)), "lib")
    subprocess.check_call([os.path.join(lib, "write"), "4", "500000", "120000"], cwd=tmp_path,
                          env=dict(os.environ, HJ_SEED="13"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cols = [np.fromfile(tmp_path / ("%s_%d.txt" % (p, n)), dtype="<u4")
            for p, n in (("ik", 120000), ("iv", 120000), ("ok", 500000), ("ov", 500000))]
    want = oracle.join_definition(*cols)
    env = dict(os.environ, HJGPU_TRANSPORT="loopback", HJGPU_RANKS="3")
    for prog in ("npj", "phj", "cpra"):
        p = subprocess.run([os.path.join(lib, prog), "8", "500000", "120000"], cwd=tmp_path, env=env,
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert "3 ranks (loopback)" in p.stderr
        assert ("join_tuples=%d sum_keys=%d sum_outer_vals=%d sum_inner_vals=%d" % want) in p.stderr
        lines = p.stdout.strip().splitlines()
        assert float(lines[-1].split("\t")[0]) > 0
        # HJGPU_ROWS on several ranks (round 3: no longer one device only): every rank materialises its share, the shares
        # arrive back to back in the host columns and are written as <prefix>jk_<J>.txt ...
        prefix = "./m3_%s_" % prog
        p = subprocess.run([os.path.join(lib, prog), "8", "500000", "120000"], cwd=tmp_path, env=dict(env, HJGPU_ROWS=prefix),
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert "3 ranks (loopback)" in p.stderr and "column sums match" in p.stderr
        got = [np.fromfile(tmp_path / ("m3_%s_%s_%d.txt" % (prog, c, want[0])), dtype="<u4") for c in ("jk", "jo", "ji")]
        for a, b in zip(sort_rows(*got), materialised_rows(*cols)):
            assert np.array_equal(a, b)
    # a device the box does not have: reported, exit code 1, no crash
    p = subprocess.run([os.path.join(lib, "phj"), "8", "500000", "120000"], cwd=tmp_path,
                       env=dict(os.environ, HJGPU_DEVICES="0,63"), capture_output=True, text=True)
    assert p.returncode == 1 and "hjgpu_comm_create_local" in p.stderr


@pytest.mark.parametrize("algorithm", [0, 1, 2])
@pytest.mark.parametrize("pinned", [False, True])
def test_join_host_rows_returns_the_materialised_join(hj, algorithm, pinned):
    """hjgpu_join_host_rows: host columns in, the dense result rows in three host
    columns out.  3.4 M result rows = more than one 32 MiB staging buffer per column on the pageable
    path; duplicates on the build side so that J != |S|."""
    rng = np.random.default_rng(77 + algorithm)
    base = np.unique(rng.integers(1, 2**32, size=200_000, dtype=np.uint64).astype(np.uint32))
    ik = np.concatenate([base, base[:50_000]])
    iv = rng.integers(0, 2**32, size=len(ik), dtype=np.uint64).astype(np.uint32)
    ok = base[rng.integers(0, len(base), size=2_700_000)]
    ov = rng.integers(0, 2**32, size=len(ok), dtype=np.uint64).astype(np.uint32)
    want = numpy_join(ik, iv, ok, ov)
    assert want[0] > 3_000_000
    got, st, rows = hj.join_host_rows(algorithm, ik, iv, ok, ov, want[0] + 12345, pinned=pinned)
    assert got == want and len(rows[0]) == want[0]
    assert st["ms_download"] > 0 and st["ms_close_gaps"] >= 0
    for a, b in zip(sort_rows(*rows), materialised_rows(ik, iv, ok, ov)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("algorithm", [0, 1, 2])
@pytest.mark.parametrize("pinned", [False, True])
def test_join_host_rows_in_batches_go_home_behind_the_upload(hj, algorithm, pinned):
    """hjgpu_join_host_rows with the probe side in batches (option "host_batch"): every batch's rows are made dense on the
    device and copied into the caller's columns while the next batch is joined; the batches' rows follow each other,
    the whole is the join row for row.  Duplicates on the build side (J != |S|); then a probe side whose matches sit
    in ONE batch (that batch outgrows its device columns and is joined once more alone, into columns made for its rows);
    then too small a capacity (the batched attempt is abandoned, the whole-column path reports HJGPU_EOVERFLOW with the
    count, as without batches)."""
    with H.HjGpu() as ctx:
        ctx.set_option("host_batch", 300_000)
        rng = np.random.default_rng(177 + algorithm)
        base = np.unique(rng.integers(1, 2**32, size=200_000, dtype=np.uint64).astype(np.uint32))
        ik = np.concatenate([base, base[:50_000]])
        iv = rng.integers(0, 2**32, size=len(ik), dtype=np.uint64).astype(np.uint32)
        ok = base[rng.integers(0, len(base), size=2_700_011)]
        ov = rng.integers(0, 2**32, size=len(ok), dtype=np.uint64).astype(np.uint32)
        want = numpy_join(ik, iv, ok, ov)
        got, st, rows = ctx.join_host_rows(algorithm, ik, iv, ok, ov, want[0] + 12345, pinned=pinned)
        assert got == want and len(rows[0]) == want[0]
        assert st["batches"] == 10 and st["ms_download"] > 0
        for a, b in zip(sort_rows(*rows), materialised_rows(ik, iv, ok, ov)):
            assert np.array_equal(a, b)
        # all matches in the first batch, none elsewhere
        ok2 = ok.copy()
        ok2[300_000:] = (ok2[300_000:] ^ np.uint32(0x5a5a5a5a)) | np.uint32(1)
        hit = np.isin(ok2, ik)
        want2 = numpy_join(ik, iv, ok2, ov)
        assert want2[0] > 0 and hit[300_000:].sum() < want2[0] // 4
        got2, st2, rows2 = ctx.join_host_rows(algorithm, ik, iv, ok2, ov, want2[0], pinned=pinned)
        assert got2 == want2 and len(rows2[0]) == want2[0]
        assert st2["batches"] == 10      # round 4: the batch that outgrew its device columns was joined once more alone, nothing started over
        for a, b in zip(sort_rows(*rows2), materialised_rows(ik, iv, ok2, ov)):
            assert np.array_equal(a, b)
        with pytest.raises(H.HjGpuError) as e:
            ctx.join_host_rows(algorithm, ik, iv, ok, ov, want[0] // 2, pinned=pinned)
        assert e.value.status == H.api.EOVERFLOW


def test_placed_allocations_for_the_callers_result_columns(hj, oracle):
    """hjgpu_malloc_placed: the workspace's placement search for a caller's buffer of a gigabyte and more (the result
    columns of a materialising join); smaller buffers are plain allocations.  The memory behaves like any other, the
    search's time shows up in ms_reserve."""
    before = hj.stats()["ms_reserve"]
    small = hj.column(1 << 20, placed=True)
    big = hj.column((1 << 28) + 4096, placed=True)              # 1 GiB + 16 KiB
    assert hj.stats()["ms_reserve"] > before
    ik, iv, ok, ov = oracle.generate(600_000, 150_000, seed=91)
    want = oracle.join_definition(ik, iv, ok, ov)
    rk, rv, sk, sv = (hj.column(c) for c in (ik, iv, ok, ov))
    cap = big.n // 4096 * 4096
    got = hj.phj(rk, rv, len(ik), sk, sv, len(ok), out=(big.ptr, big.ptr + 4 * (cap // 3 // 4096 * 4096), big.ptr + 8 * (cap // 3 // 4096 * 4096), cap // 3 // 4096 * 4096, 4096))
    assert got == want
    assert hj.column_sums(big.ptr, want[0], 1, 1)[0] == want[1]
    for c in (small, big, rk, rv, sk, sv):
        c.free()


def test_join_host_rows_reports_overflow_with_the_row_count(hj, oracle):
    """More result rows than rows->capacity: HJGPU_EOVERFLOW, and the caller can size a retry."""
    ik, iv, ok, ov = oracle.generate(300_000, 60_000, seed=46)
    want = oracle.join_definition(ik, iv, ok, ov)
    with pytest.raises(H.HjGpuError) as e:
        hj.join_host_rows(1, ik, iv, ok, ov, want[0] // 2)
    assert e.value.status == 6
    got, _, rows = hj.join_host_rows(1, ik, iv, ok, ov, want[0])      # exactly enough
    assert got == want and len(rows[0]) == want[0]
    # an empty join fills nothing
    got, _, rows = hj.join_host_rows(2, ik, iv, (ok ^ np.uint32(0x5a5a5a5a)) | np.uint32(1), ov, 16)
    if got[0] == 0:
        assert len(rows[0]) == 0


def test_host_programs_materialise_to_host_columns(hj, oracle, tmp_path):
    """HJGPU_ROWS=<prefix>: the host programs return the join as three columns in host memory (what
    the reference's mains hold after run(), npj.cpp:997-1000) and write them as raw uint32 files."""
    import subprocess
    lib = os.path.join(os.path.dirname(os.path.abspath(H.__file__)), "lib")
    subprocess.check_call([os.path.join(lib, "write"), "4", "300000", "70000"], cwd=tmp_path,
                          env=dict(os.environ, HJ_SEED="12"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cols = [np.fromfile(tmp_path / ("%s_%d.txt" % (p, n)), dtype="<u4")
            for p, n in (("ik", 70000), ("iv", 70000), ("ok", 300000), ("ov", 300000))]
    want = numpy_join(*cols)
    for prog in ("npj", "phj", "cpra"):
        prefix = "./%s_" % prog
        p = subprocess.run([os.path.join(lib, prog), "8", "300000", "70000"], cwd=tmp_path,
                           env=dict(os.environ, HJGPU_ROWS=prefix), capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert "column sums match" in p.stderr
        got = [np.fromfile(tmp_path / ("%s_%s_%d.txt" % (prog, c, want[0])), dtype="<u4") for c in ("jk", "jo", "ji")]
        for a, b in zip(sort_rows(*got), materialised_rows(*cols)):
            assert np.array_equal(a, b)


def test_prepared_build_probed_in_batches(hj):
    """hjgpu_phj_build + hjgpu_phj_probe: the build side is partitioned once, batches of the probe side are
    joined against it; the batch results add up to the whole join's (R join S = union of R join S_i), for a
    two-pass and a single-pass plan, ragged batch sizes, an empty batch, and materialised rows per batch."""
    rng = np.random.default_rng(321)
    base = np.unique(rng.integers(0, 2**32, size=700_000, dtype=np.uint64).astype(np.uint32))
    ik = np.concatenate([base, base[:90_000]])
    iv = rng.integers(0, 2**32, size=len(ik), dtype=np.uint64).astype(np.uint32)
    ok = np.where(rng.random(3_000_000) < 0.7, base[rng.integers(0, len(base), size=3_000_000)],
                  rng.integers(0, 2**32, size=3_000_000, dtype=np.uint64).astype(np.uint32)).astype(np.uint32)
    ov = rng.integers(0, 2**32, size=len(ok), dtype=np.uint64).astype(np.uint32)
    want = numpy_join(ik, iv, ok, ov)
    rk, rv, sk, sv = _cols(hj, ik, iv, ok, ov)
    mask = (1 << 64) - 1
    for prm in (None, H.PhjParams(fanout1=37, fanout2=1), H.PhjParams(fanout1=64, fanout2=40)):
        cuts = [0, 16 * 3, 16 * 40_000, 16 * 40_000, 16 * 100_001, len(ok)]        # batches start on 64-byte boundaries
        hj.phj_build(rk, rv, len(ik), max(b - a for a, b in zip(cuts, cuts[1:])), prm)
        total = (0, 0, 0, 0)
        for a, b in zip(cuts, cuts[1:]):
            got = hj.phj_probe(sk.ptr + 4 * a, sv.ptr + 4 * a, b - a)
            assert got == numpy_join(ik, iv, ok[a:b], ov[a:b])
            total = tuple((x + y) & mask for x, y in zip(total, got))
        assert total == want
    # rows of one batch
    a, b = 16 * 1000, 16 * 9000
    w = numpy_join(ik, iv, ok[a:b], ov[a:b])
    block = 1024
    cap = (w[0] // block + hj.device_info()["compute_units"] * 16 + 8) * block
    jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
    assert hj.phj_probe(sk.ptr + 4 * a, sv.ptr + 4 * a, b - a, out=(jk, jo, ji, cap, block)) == w
    rows = sort_rows(jk.download()[:w[0]], jo.download()[:w[0]], ji.download()[:w[0]])
    for x, y in zip(rows, materialised_rows(ik, iv, ok[a:b], ov[a:b])):
        assert np.array_equal(x, y)
    # a batch beyond max_outer is refused; another operator on the context ends the prepared state
    with pytest.raises(H.HjGpuError) as e:
        hj.phj_probe(sk, sv, len(ok))
    assert e.value.status == 1
    assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == want
    with pytest.raises(H.HjGpuError) as e:
        hj.phj_probe(sk, sv, 1024)
    assert e.value.status == 1
    _free(rk, rv, sk, sv, jk, jo, ji)


def test_async_joins_back_to_back_on_a_side_stream_and_no_graph_capture(hj, oracle):
    """Twenty hjgpu_phj_async joins enqueued on a non-default stream without host synchronisation in between
    (every join re-zeroes its histograms, tickets and result block in stream order); a capturing stream is
    refused with HJGPU_EINVAL (a replayed graph of a join faulted on gfx950 / ROCm 7.0, include/hjgpu.h)."""
    import torch
    ik, iv, ok, ov = oracle.generate(700_000, 150_000, seed=47)
    want = oracle.join_definition(ik, iv, ok, ov)
    rk, rv, sk, sv = _cols(hj, ik, iv, ok, ov)
    hj.reserve(len(ik), len(ok))
    d_res = torch.zeros(4, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    for _ in range(20):
        hj.phj_async(rk, rv, len(ik), sk, sv, len(ok), None, d_res.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert tuple(int(x) & ((1 << 64) - 1) for x in d_res.tolist()) == want
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        g.capture_begin()
        try:
            for fn, prm in ((hj.phj_async, None), (hj.npj_async, None), (hj.cpra_async, H.PhjParams(chunks=2))):
                with pytest.raises(H.HjGpuError) as e:
                    fn(rk, rv, len(ik), sk, sv, len(ok), prm, d_res.data_ptr(), s.cuda_stream)
                assert e.value.status == 1
        finally:
            g.capture_end()
    s.synchronize()
    assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == want       # the context is still usable
    _free(rk, rv, sk, sv)
