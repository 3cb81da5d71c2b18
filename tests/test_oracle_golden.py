"""CPU tests: the oracle (oracle/hj_oracle.c) against the golden vectors produced by
the REFERENCE'S OWN scalar operator code (tests/golden/make_golden.py), and -
when oracle/_ref is present (authoring container) - directly against it."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from helpers import numpy_join, pairs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HIST_CASES = [(0x9E3779B1, 7), (0x85EBCA6B, 64), (0x9E3779B1, 1000)]
PART_CASES = [(0x9E3779B1, 7), (0x85EBCA6B, 64)]
NPJ_FACTOR, NPJ_LOAD = 0x9E3779B1, 0.90
PHJ_FACTORS = (0xC2B2AE35, 0x27D4EB2F)
FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                  if not os.path.basename(p).startswith("rand32"))


@pytest.fixture(params=["scalar", "avx512", "avx512_vector_scatter"], autouse=True)
def operator_forms(request, oracle):
    """Every test of this module runs three times: with the scalar definitions (the oracle proper), with the
    AVX-512 forms of histogram / partition / probe that bench.py times as the CPU baseline, and with the
    partition's conflict-serialised vector scatter (the reference's shape, phj.cpp:1099-1160).  All must
    reproduce the reference's outputs bit for bit."""
    if request.param != "scalar":
        if not oracle.simd_available():
            pytest.skip("no AVX-512 on this CPU")
        want = 2 if request.param == "avx512_vector_scatter" else 1
        assert oracle.set_simd(want) == want
    yield request.param
    oracle.set_simd(False)


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def test_fixtures_present():
    assert len(FIXTURES) >= 4


def test_rand32_stream_matches_reference(oracle):
    class RS(C.Structure):
        _fields_ = [("num", C.c_uint32 * 625), ("index", C.c_size_t)]
    want = np.load(os.path.join(GOLDEN, "rand32_seed5489.npz"))["stream"]
    s = RS()
    L = oracle.lib()
    L.hjo_rand32_init(C.byref(s), 5489)
    L.hjo_rand32_next.restype = C.c_uint32
    got = np.array([L.hjo_rand32_next(C.byref(s)) for _ in range(len(want))], np.uint32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name", [f for f in FIXTURES if f not in ("key_zero_and_extremes", "dups4_distinct_payloads")])
def test_generator_reproduces_reference_relations(oracle, name):
    g = load(name)
    outer, inner, sel1000, seed, uf, fi, fo = (int(x) for x in g["gen_params"])
    ik, iv, ok, ov = oracle.generate(outer, inner, selectivity=sel1000 / 1000.0, seed=seed,
                                     unique_factor=uf, inner_factor=fi, outer_factor=fo)
    for a, b in ((ik, g["inner_keys"]), (iv, g["inner_vals"]), (ok, g["outer_keys"]), (ov, g["outer_vals"])):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", FIXTURES)
def test_histogram_and_partition(oracle, name):
    g = load(name)
    ok, ov = g["outer_keys"], g["outer_vals"]
    for idx, (f, F) in enumerate(HIST_CASES):
        assert np.array_equal(oracle.histogram(ok, f, F), g["hist_%d" % idx])
    for idx, (f, F) in enumerate(PART_CASES):
        counts, ko, vo = oracle.partition(ok, ov, f, F)
        assert np.array_equal(counts, g["part_%d_counts" % idx])
        off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
        for p in range(F):
            assert int(ko[off[p]:off[p + 1]].astype(np.uint64).sum()) == int(g["part_%d_sum_keys" % idx][p])
            assert int(vo[off[p]:off[p + 1]].astype(np.uint64).sum()) == int(g["part_%d_sum_vals" % idx][p])
        if idx == 0:       # both are stable counting sorts: element-wise equal
            assert np.array_equal(ko, g["part_0_keys"]) and np.array_equal(vo, g["part_0_vals"])


@pytest.mark.parametrize("name", FIXTURES)
def test_join_operators(oracle, name):
    g = load(name)
    ik, iv, ok, ov = g["inner_keys"], g["inner_vals"], g["outer_keys"], g["outer_vals"]
    L = oracle.lib()
    want_def = numpy_join(ik, iv, ok, ov)
    # PHJ build/probe on one partition with the reference's bucket count and sentinel
    buckets, empty = (int(x) for x in g["phj_buckets"])
    table = np.zeros(buckets, np.uint64)
    fac = (C.c_uint32 * 2)(*PHJ_FACTORS)
    L.hjo_phj_build(ik, iv, len(ik), table, buckets, fac, empty)
    assert np.array_equal(np.sort(table), g["phj_table_sorted"])
    r = oracle.Result()
    L.hjo_phj_probe(ok, ov, len(ok), table, buckets, fac, empty, C.byref(r), None, None)
    assert r.as_tuple() == tuple(int(x) for x in g["phj_result"]) == want_def
    if "npj_result" in g:
        buckets = int(len(ik) / NPJ_LOAD)
        table = np.zeros(buckets, np.uint64)
        L.hjo_npj_build(ik, iv, len(ik), table, buckets, NPJ_FACTOR, 0)
        assert np.array_equal(np.sort(table), g["npj_table_sorted"])
        r = oracle.Result()
        L.hjo_npj_probe(ok, ov, len(ok), table, buckets, NPJ_FACTOR, 0, C.byref(r), None, None)
        assert r.as_tuple() == tuple(int(x) for x in g["npj_result"]) == want_def
        # whole joins, several thread counts, all equal the reference's aggregates
        for T in (1, 2, 5):
            assert oracle.npj(ik, iv, ok, ov, threads=T, load=NPJ_LOAD, factor=NPJ_FACTOR) == want_def
            res, (jk, jo, ji) = oracle.npj(ik, iv, ok, ov, threads=T, materialize=True, block_size=256)
            assert res == want_def and len(jk) == want_def[0]
            assert int(jk.astype(np.uint64).sum()) == want_def[1]
    for T in (1, 2, 4):
        assert oracle.phj(ik, iv, ok, ov, threads=T, hash_table_limit=64) == want_def
        assert oracle.cpra(ik, iv, ok, ov, threads=T, num_partitions=64) == want_def


@pytest.mark.parametrize("name", FIXTURES)
def test_unique_probes_match_the_reference_built_with_UNIQUE(oracle, name):
    """hjo_set_unique(1) against the outputs of the reference's own probe / probe_s compiled with -D_UNIQUE
    (npj.cpp:436-438, phj.cpp:635-637).  dups4_distinct_payloads gives a key's duplicates different payloads:
    its sum_inner pins WHICH duplicate is first (at one thread: the one inserted first)."""
    g = load(name)
    ik, iv, ok, ov = g["inner_keys"], g["inner_vals"], g["outer_keys"], g["outer_vals"]
    L = oracle.lib()
    assert oracle.set_unique(True)
    try:
        buckets, empty = (int(x) for x in g["phj_buckets"])
        table = np.zeros(buckets, np.uint64)
        fac = (C.c_uint32 * 2)(*PHJ_FACTORS)
        L.hjo_phj_build(ik, iv, len(ik), table, buckets, fac, empty)
        r = oracle.Result()
        L.hjo_phj_probe(ok, ov, len(ok), table, buckets, fac, empty, C.byref(r), None, None)
        want = tuple(int(x) for x in g["phj_result_unique"])
        assert r.as_tuple() == want
        assert want[:3] == oracle.join_definition_unique(ik, iv, ok, ov)
        if "npj_result_unique" in g:
            buckets = int(len(ik) / NPJ_LOAD)
            table = np.zeros(buckets, np.uint64)
            L.hjo_npj_build(ik, iv, len(ik), table, buckets, NPJ_FACTOR, 0)
            r = oracle.Result()
            L.hjo_npj_probe(ok, ov, len(ok), table, buckets, NPJ_FACTOR, 0, C.byref(r), None, None)
            assert r.as_tuple() == tuple(int(x) for x in g["npj_result_unique"])
            # whole joins: count and the probe-side sums do not depend on the plan or the thread count
            for T in (1, 3):
                assert oracle.npj(ik, iv, ok, ov, threads=T, load=NPJ_LOAD, factor=NPJ_FACTOR)[:3] == want[:3]
                assert oracle.phj(ik, iv, ok, ov, threads=T, hash_table_limit=64)[:3] == want[:3]
                assert oracle.cpra(ik, iv, ok, ov, threads=T, num_partitions=64)[:3] == want[:3]
    finally:
        oracle.set_unique(False)


def test_oracle_primitives(oracle):
    L = oracle.lib()
    assert L.hjo_hash(0xFFFFFFFF, 10) == 9 and L.hjo_hash(0, 12345) == 0 and L.hjo_hash(0x80000000, 2) == 1
    # npj.cpp:516-529
    assert [L.hjo_thread_beg(1000, 16, t, 3) for t in range(3)] == [0, 320, 640]
    assert [L.hjo_thread_end(1000, 16, t, 3) for t in range(3)] == [320, 640, 1000]
    assert L.hjo_next_odd_prime(16000) == 16001 and L.hjo_next_odd_prime(8) == 11
    assert L.hjo_odd_prime(9) == 0 and L.hjo_odd_prime(16001) == 1


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_matches_reference_build_when_present(oracle, seed):
    """Randomised cross-check against oracle/_ref (only where it has been built)."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    R = oracle.ref()
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1000, 50000))
    keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    vals = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    F = int(rng.integers(2, 3000))
    f = int(rng.integers(0, 2**31)) * 2 + 1
    c = np.zeros(F, np.uint32)
    R.hjref_histogram(keys, n, c, f, F)
    assert np.array_equal(c, oracle.histogram(keys, f, F))
    b = np.zeros(n + 32, np.uint32); o = (-b.ctypes.data % 64) // 4; ko = b[o:o + n]
    b2 = np.zeros(n + 32, np.uint32); o2 = (-b2.ctypes.data % 64) // 4; vo = b2[o2:o2 + n]
    R.hjref_partition(keys, vals, n, c, ko, vo, f, F)
    _, ko2, vo2 = oracle.partition(keys, vals, f, F)
    assert np.array_equal(ko, ko2) and np.array_equal(vo, vo2)
    for T in (1, 3, 7):
        for t in range(T):
            assert R.hjref_thread_beg(n, 16, t, T) == oracle.lib().hjo_thread_beg(n, 16, t, T)
            assert R.hjref_thread_end(n, 16, t, T) == oracle.lib().hjo_thread_end(n, 16, t, T)


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_whole_joins_agree_with_the_join_definition(oracle, threads):
    """run / run_hj restatements (thread-level pass with ragged thread ranges, local passes,
    per-partition build + probe) against the sort-merge definition of the join, in both
    operator forms (the AVX-512 partition must cope with output ranges that start and end in
    the middle of a 64-byte line and are shared with other threads)."""
    ik, iv, ok, ov = oracle.generate(300_007, 61_003, seed=threads)
    want = oracle.join_definition(ik, iv, ok, ov)
    assert oracle.phj(ik, iv, ok, ov, threads=threads, hash_table_limit=700) == want
    assert oracle.cpra(ik, iv, ok, ov, threads=threads) == want
    assert oracle.npj(ik, iv, ok, ov, threads=threads) == want
