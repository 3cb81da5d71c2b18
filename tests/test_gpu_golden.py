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
    os.environ["HJGPU_FORCE_CHAINED"] = "1"
    try:
        assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == want
        assert hj.cpra(rk, rv, len(ik), sk, sv, len(ok), H.PhjParams(fanout1=7, fanout2=2, chunks=3)) == want
    finally:
        del os.environ["HJGPU_FORCE_CHAINED"]
    for c in (rk, rv, sk, sv):
        c.free()


def test_full_size_properties_64m_1g(hj):
    """BASELINE.json configs[1]/[2] sizes: |R| = 64 M, |S| = 1 G, selectivity 1.
    Size-independent properties: (i) join count = |S| and the three sums equal the
    column checksums of S (every probe key matches exactly one build key);
    (ii) partitioning preserves the column checksums (linearity) and every sampled
    partition range holds only keys of that partition; (iii) NPJ, PHJ and CPRA agree."""
    inner, outer = 64_000_000, 1_000_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(1, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, fo, fi)
    want = (outer, sums[0], sums[1], sums[2])
    hj.reserve(inner, outer)
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    st = hj.stats()
    assert st["fanout1"] * st["fanout2"] >= 15_000
    assert hj.cpra(ik, iv, inner, ok, ov, outer, H.PhjParams(chunks=8)) == want
    assert hj.npj(ik, iv, inner, ok, ov, outer) == want
    # (ii) one pass over S with fan-out 1000
    F, f = 1000, 0x85EBCA6B
    pk, pv, off = hj.column(outer), hj.column(outer), hj.column(F + 1, np.uint64)
    hj.partition(ok, ov, outer, f, F, pk, pv, off)
    assert hj.column_sums(pk, outer, fo, fi) == sums
    o = off.download().astype(np.int64)
    assert o[0] == 0 and o[-1] == outer and (np.diff(o) > 0).all()
    cnt = hj.column(F, np.uint64)
    for p in (0, 1, 499, 998, 999):
        n = int(o[p + 1] - o[p])
        hj.histogram(pk.ptr + 4 * int(o[p]), n, f, F, cnt)
        c = cnt.download()
        assert c[p] == n and c.sum() == n
    for c in (ik, iv, ok, ov, pk, pv, off, cnt):
        c.free()


def test_partition_sample_is_in_partition(hj):
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 2**32, size=5_000_000, dtype=np.uint64).astype(np.uint32)
    dk, dv = hj.column(keys), hj.column(keys)
    pk, pv, off = hj.column(len(keys)), hj.column(len(keys)), hj.column(513, np.uint64)
    hj.partition(dk, dv, len(keys), 0x9E3779B1, 512, pk, pv, off)
    o = off.download().astype(np.int64)
    ko = pk.download()
    assert np.array_equal(mulhi_hash(ko, 0x9E3779B1, 512), np.searchsorted(o, np.arange(len(keys)), side="right") - 1)
    assert np.array_equal(pv.download(), ko)
    for c in (dk, dv, pk, pv, off):
        c.free()


def test_probe_side_beyond_2_32_tuples(hj):
    """the reference's uint32 offsets cannot address >= 2^32 tuples per
    relation; this library uses 64-bit offsets everywhere.  |S| = 4.4 G (> 2^32) probe
    tuples against |R| = 16 M: count and the three sums must equal the column checksums."""
    inner, outer = 16_000_000, 4_400_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(9, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, fo, fi)
    want = (outer, sums[0], sums[1], sums[2])
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    # the tail beyond element 2^32 really took part: joining only the first 2^32 tuples differs
    head = hj.phj(ik, iv, inner, ok, ov, 1 << 32)
    assert head[0] == 1 << 32 and head != want
    tail_n = outer - (1 << 32)
    tail = hj.phj(ik, iv, inner, ok.ptr + 4 * (1 << 32), ov.ptr + 4 * (1 << 32), tail_n)
    assert tail[0] == tail_n
    assert tuple((a + b) & ((1 << 64) - 1) for a, b in zip(head, tail)) == want
    for c in (ik, iv, ok, ov):
        c.free()


def test_beyond_2_31_tuples_128m_2g2(hj):
    """BASELINE.json configs[4]'s per-GPU shape (CPRA, |R| = 1 G / 8 GPUs, |S| = 16 G / 8 GPUs): a build
    side of 128 M tuples, for which the library switches to 16 K-slot LDS tables (8 K-slot ones would need
    more than HJGPU_MAX_PARTS partitions), and a probe side of more than 2^31 tuples, so every offset,
    cursor and output slot index above 32 bits is exercised.  Properties as in the 64 M x 1 G test, plus
    the materialised result: J = |S| rows, dense, whose column sums are the aggregates."""
    inner, outer = 128_000_000, 2_200_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(3, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, fo, fi)
    want = (outer, sums[0], sums[1], sums[2])
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    st = hj.stats()
    assert 16_000 <= st["fanout1"] * st["fanout2"] <= 32768
    assert hj.cpra(ik, iv, inner, ok, ov, outer, H.PhjParams(chunks=8)) == want
    assert hj.npj(ik, iv, inner, ok, ov, outer) == want
    block = 65536
    cap = (outer // block + hj.device_info()["compute_units"] * 16 + 8) * block
    jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
    assert hj.phj(ik, iv, inner, ok, ov, outer, out=(jk, jo, ji, cap, block)) == want
    assert hj.column_sums(jk, outer, 1, 1)[0] == want[1]
    assert hj.column_sums(jo, outer, 1, 1)[0] == want[2]
    assert hj.column_sums(ji, outer, 1, 1)[0] == want[3]
    for c in (ik, iv, ok, ov, jk, jo, ji):
        c.free()


def test_beyond_2_32_probe_tuples_64m_4g4(hj):
    """A probe side of 4.4 G tuples (> 2^32; 35 GB of columns, 70 GB of scratch twins: what one 288 GB GPU
    holds): every tuple index, tile count and offset beyond 32 bits.  count = |S| shows that every probe key
    found its build key; the sums are the column checksums of S."""
    inner, outer = 64_000_000, 4_400_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
    hj.generate(5, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
    sums = hj.column_sums(ok, outer, fo, fi)
    want = (outer, sums[0], sums[1], sums[2])
    assert hj.phj(ik, iv, inner, ok, ov, outer) == want
    assert hj.npj(ik, iv, inner, ok, ov, outer) == want
    for c in (ik, iv, ok, ov):
        c.free()


def test_config1_npj_1m_probe_16m_build(hj, oracle):
    """BASELINE.json configs[0], `./npj 64 1000000 16000000`: outer = 1 M probe tuples, inner = 16 M build
    tuples over 1 M distinct keys (16 copies per key, write.cpp semantics), J = 16 M.  All three algorithms
    against the independent numpy definition, aggregates and the dense materialised row count."""
    ik, iv, ok, ov = oracle.generate(1_000_000, 16_000_000, seed=1)
    assert len(ik) == 16_000_000 and len(ok) == 1_000_000
    want = numpy_join(ik, iv, ok, ov)
    assert 15_000_000 < want[0] < 17_000_000
    rk, rv, sk, sv = (hj.column(c) for c in (ik, iv, ok, ov))
    assert hj.npj(rk, rv, len(ik), sk, sv, len(ok), H.NpjParams(load=0.9)) == want      # npj.cpp:944 load factor
    assert hj.npj(rk, rv, len(ik), sk, sv, len(ok)) == want
    assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == want
    assert hj.cpra(rk, rv, len(ik), sk, sv, len(ok), H.PhjParams(chunks=8)) == want
    block = 65536
    cap = (want[0] // block + hj.device_info()["compute_units"] * 32 + 8) * block
    jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
    assert hj.npj(rk, rv, len(ik), sk, sv, len(ok), out=(jk, jo, ji, cap, block)) == want
    assert hj.column_sums(jk, want[0], 1, 1)[0] == want[1]
    assert hj.column_sums(jo, want[0], 1, 1)[0] == want[2]
    assert hj.column_sums(ji, want[0], 1, 1)[0] == want[3]
    for c in (rk, rv, sk, sv, jk, jo, ji):
        c.free()
