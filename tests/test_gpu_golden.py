"""GPU tests against the committed golden vectors (outputs of the reference's own
scalar operator code, tests/golden/make_golden.py) and size-independent
properties at BASELINE.json's full sizes."""
import glob
import os

import numpy as np
import pytest

import hash_join_codes_knl_amd as H
from helpers import mulhi_hash, numpy_join

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HIST_CASES = [(0x9E3779B1, 7), (0x85EBCA6B, 64), (0x9E3779B1, 1000)]
PART_CASES = [(0x9E3779B1, 7), (0x85EBCA6B, 64)]
NPJ_FACTOR, NPJ_LOAD = 0x9E3779B1, 0.90
FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                  if not os.path.basename(p).startswith("rand32"))


@pytest.mark.parametrize("name", FIXTURES)
def test_hip_operators_reproduce_reference_outputs(hj, name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ik, iv, ok, ov = g["inner_keys"], g["inner_vals"], g["outer_keys"], g["outer_vals"]
    rk, rv, sk, sv = hj.column(ik), hj.column(iv), hj.column(ok), hj.column(ov)
    # K4
    for idx, (f, F) in enumerate(HIST_CASES):
        dc = hj.column(F, np.uint64)
        hj.histogram(sk, len(ok), f, F, dc)
        assert np.array_equal(dc.download(), g["hist_%d" % idx].astype(np.uint64))
        dc.free()
    # K5 + K6
    for idx, (f, F) in enumerate(PART_CASES):
        dko, dvo, doff = hj.column(len(ok)), hj.column(len(ok)), hj.column(F + 1, np.uint64)
        hj.partition(sk, sv, len(ok), f, F, dko, dvo, doff)
        off = doff.download().astype(np.int64)
        ko, vo = dko.download(), dvo.download()
        assert np.array_equal(np.diff(off), g["part_%d_counts" % idx].astype(np.int64))
        for p in range(F):
            assert int(ko[off[p]:off[p + 1]].astype(np.uint64).sum()) == int(g["part_%d_sum_keys" % idx][p])
            assert int(vo[off[p]:off[p + 1]].astype(np.uint64).sum()) == int(g["part_%d_sum_vals" % idx][p])
        for c in (dko, dvo, doff):
            c.free()
    want = tuple(int(x) for x in g["phj_result"])
    # K1-K3 with the reference's load factor and hash factor: same bucket multiset, same result
    if "npj_result" in g:
        buckets = int(len(ik) / NPJ_LOAD)
        dt = hj.column(buckets, np.uint64)
        hj.npj_build(rk, rv, len(ik), dt, buckets, NPJ_FACTOR)
        assert np.array_equal(np.sort(dt.download()), g["npj_table_sorted"])
        assert hj.npj_probe(sk, sv, len(ok), dt, buckets, NPJ_FACTOR) == tuple(int(x) for x in g["npj_result"])
        dt.free()
        assert hj.npj(rk, rv, len(ik), sk, sv, len(ok), H.NpjParams(load=NPJ_LOAD, factor=NPJ_FACTOR)) == want
    # K7 + K8 (cuckoo fast path, then the chained fallback forced everywhere)
    for prm in (None, H.PhjParams(fanout1=5, fanout2=3), H.PhjParams(fanout1=64, fanout2=1, chunks=4)):
        assert hj.phj(rk, rv, len(ik), sk, sv, len(ok), prm) == want
        assert hj.cpra(rk, rv, len(ik), sk, sv, len(ok), prm) == want
    hj.set_option("force_chained", 1)
    try:
        assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == want
        assert hj.cpra(rk, rv, len(ik), sk, sv, len(ok), H.PhjParams(fanout1=7, fanout2=2, chunks=3)) == want
    finally:
        hj.set_option("force_chained", 0)
    for c in (rk, rv, sk, sv):
        c.free()


def _rows_are_valid_unique_result(ik, iv, ok, ov, rows):
    """A _UNIQUE result: exactly one row per probe tuple that has a match, carrying the payload of ONE of
    the build tuples with its key (which one is unspecified under parallel insertion)."""
    jk, jo, ji = rows
    hit = np.isin(ok, ik)
    want = np.sort((ok[hit].astype(np.uint64) << np.uint64(32)) | ov[hit])
    got = np.sort((jk.astype(np.uint64) << np.uint64(32)) | jo)
    assert np.array_equal(got, want)
    build = np.unique((ik.astype(np.uint64) << np.uint64(32)) | iv)
    assert np.isin((jk.astype(np.uint64) << np.uint64(32)) | ji, build).all()


@pytest.mark.parametrize("name", FIXTURES)
def test_unique_mode_reproduces_the_reference_built_with_UNIQUE(hj, name):
    """HJGPU_FLAG_UNIQUE / option "unique" against the reference's probe / probe_s compiled with -D_UNIQUE
    (npj.cpp:436-438, phj.cpp:635-637): every algorithm, cuckoo and chained tables, one- and two-pass plans,
    a plan whose partitions take several table fills, operator-level entry points, materialised rows."""
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ik, iv, ok, ov = g["inner_keys"], g["inner_vals"], g["outer_keys"], g["outer_vals"]
    want = tuple(int(x) for x in g["phj_result_unique"])
    # duplicates with different payloads: the build payload is ONE of the key's, which one depends on insertion order
    exact = name != "dups4_distinct_payloads"
    rk, rv, sk, sv = hj.column(ik), hj.column(iv), hj.column(ok), hj.column(ov)
    U = H.FLAG_UNIQUE

    def check(got):
        assert got[:3] == want[:3]
        if exact:
            assert got == want

    plans = (H.PhjParams(flags=U), H.PhjParams(fanout1=5, fanout2=3, flags=U), H.PhjParams(fanout1=64, fanout2=1, chunks=4, flags=U),
             H.PhjParams(fanout1=2, fanout2=1, chunks=2, flags=U))
    for chained in (0, 1):
        hj.set_option("force_chained", chained)
        try:
            for prm in plans:
                check(hj.phj(rk, rv, len(ik), sk, sv, len(ok), prm))
                check(hj.cpra(rk, rv, len(ik), sk, sv, len(ok), prm))
        finally:
            hj.set_option("force_chained", 0)
    cap = (len(ok) // 256 + 4200) * 256
    jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
    for prm in (plans[0], plans[3]):
        got = hj.phj(rk, rv, len(ik), sk, sv, len(ok), prm, out=(jk, jo, ji, cap, 256))
        check(got)
        _rows_are_valid_unique_result(ik, iv, ok, ov, tuple(c.download(got[0]) for c in (jk, jo, ji)))
    if "npj_result_unique" in g:
        for load in (0.25, 0.9):
            check(hj.npj(rk, rv, len(ik), sk, sv, len(ok), H.NpjParams(load=load, factor=NPJ_FACTOR, flags=U)))
        got = hj.npj(rk, rv, len(ik), sk, sv, len(ok), H.NpjParams(flags=U), out=(jk, jo, ji, cap, 256))
        check(got)
        _rows_are_valid_unique_result(ik, iv, ok, ov, tuple(c.download(got[0]) for c in (jk, jo, ji)))
        # operator level: the reference's table format, probed with the context option
        buckets = int(len(ik) / NPJ_LOAD)
        dt = hj.column(buckets, np.uint64)
        hj.npj_build(rk, rv, len(ik), dt, buckets, NPJ_FACTOR)
        hj.set_option("unique", 1)
        try:
            check(hj.npj_probe(sk, sv, len(ok), dt, buckets, NPJ_FACTOR))
            check(hj.phj(rk, rv, len(ik), sk, sv, len(ok)))          # the option alone, no flag
        finally:
            hj.set_option("unique", 0)
        assert hj.npj_probe(sk, sv, len(ok), dt, buckets, NPJ_FACTOR) == tuple(int(x) for x in g["npj_result"])
        dt.free()
    for c in (rk, rv, sk, sv, jk, jo, ji):
        c.free()


def test_unique_mode_with_duplicate_heavy_build_sides(hj, oracle):
    """`_UNIQUE` on build sides with many copies per key - BASELINE configs[0]'s shape (16 copies per
    key), and 300 copies per key, where one partition takes several table fills and `matched` bits keep a probe
    row from being reported once per fill.  Broadcast-sized build sides too (<= 6963 rows, nothing partitioned).
    Every copy of a key carries payload key * factor, so sum_inner_vals is determined as well."""
    U = H.FLAG_UNIQUE
    rng = np.random.default_rng(8)
    FI, FO = 0x2545F491, 0x85EBCA6B
    cases = []
    ik, _, ok, _ = oracle.generate(200_000, 3_200_000, seed=4)                   # 16 copies per key
    cases.append((ik, ok))
    base = np.unique(rng.integers(1, 2**32, size=1200, dtype=np.uint64).astype(np.uint32))[:1000]
    ik = np.repeat(base, 300); rng.shuffle(ik)                                    # 300 copies: several fills per partition
    ok = np.concatenate([base[rng.integers(0, 1000, size=150_000)],
                         rng.integers(1, 2**32, size=50_000, dtype=np.uint64).astype(np.uint32)])
    cases.append((ik, ok))
    ik = np.repeat(base[:500], 9); rng.shuffle(ik)                                # 4500 rows: broadcast join
    cases.append((ik, ok))
    for ik, ok in cases:
        iv, ov = ik * np.uint32(FI), ok * np.uint32(FO)
        hit = np.isin(ok, ik)
        want = oracle.join_definition_unique(ik, iv, ok, ov) + (int((ok[hit] * np.uint32(FI)).astype(np.uint64).sum()),)
        rk, rv, sk, sv = (hj.column(c) for c in (ik, iv, ok, ov))
        for prm in (H.PhjParams(flags=U), H.PhjParams(fanout1=3, fanout2=1, flags=U),
                    H.PhjParams(fanout1=16, fanout2=4, chunks=3, flags=U)):
            assert hj.phj(rk, rv, len(ik), sk, sv, len(ok), prm) == want
            assert hj.cpra(rk, rv, len(ik), sk, sv, len(ok), prm) == want
        assert hj.npj(rk, rv, len(ik), sk, sv, len(ok), H.NpjParams(flags=U)) == want
        # without the flag every copy is reported
        assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == numpy_join(ik, iv, ok, ov)
        for c in (rk, rv, sk, sv):
            c.free()


def test_full_size_64m_1g_equals_the_cpu_oracle(hj, oracle):
    """The headline workload checked against the ORACLE itself, not only against aggregates the library computed:
    the device-generated 64 M x 1 G relations are copied to the host and joined by oracle/hj_oracle.c's restatement of
    run_hj (all usable threads, AVX-512 operators where present); PHJ, CPRA and NPJ on the GPU must return the same
    four aggregates.  (bench.py's cpu_baseline leg makes the same comparison in every driver run.)"""
    inner, outer = 64_000_000, 1_000_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(17, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    hik, hiv, hok, hov = ik.download(), iv.download(), ok.download(), ov.download()
    threads = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            threads = min(threads, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    oracle.set_simd(1)
    try:
        want = oracle.phj(hik, hiv, hok, hov, threads=threads)
    finally:
        oracle.set_simd(0)
    del hik, hiv, hok, hov
    assert want[0] == outer
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    assert hj.cpra(ik, iv, inner, ok, ov, outer, H.PhjParams(chunks=8)) == want
    assert hj.npj(ik, iv, inner, ok, ov, outer) == want
    # the reference's -D_UNIQUE build (phj.cpp:635-637, npj.cpp:288-290): with unique build keys the first match is the only
    # one, so the two-launch _UNIQUE join (round 4) and the _UNIQUE NPJ walk must return the same aggregates at full size
    assert hj.phj(ik, iv, inner, ok, ov, outer, H.PhjParams(flags=H.FLAG_UNIQUE)) == want
    assert hj.cpra(ik, iv, inner, ok, ov, outer, H.PhjParams(chunks=8, flags=H.FLAG_UNIQUE)) == want
    assert hj.npj(ik, iv, inner, ok, ov, outer, H.NpjParams(flags=H.FLAG_UNIQUE)) == want
    for c in (ik, iv, ok, ov):
        c.free()
