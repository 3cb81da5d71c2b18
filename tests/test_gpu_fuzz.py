"""Randomised differential tests on the GPU: shapes, fan-outs, chunk counts, duplicates, skew,
selectivities and output modes drawn from a seeded generator, every result compared bit for bit
with an independent numpy definition of the join (helpers.numpy_join / materialised_rows).

The fan-outs are chosen so that every geometry of K6 runs: whole-line mode with 16 K-, 12 K- and
8 K-tuple tiles (fan-out <= 209 / <= 421 / <= 640), no carry beyond that, single-pass plans, runs
longer than one stream-out unit (tiny fan-outs, heavy hitters), ranges without tiles (small inputs)."""
import os

import numpy as np
import pytest

import hash_join_codes_knl_amd as H
from helpers import numpy_join, materialised_rows, sort_rows

pytestmark = pytest.mark.gpu

FANOUTS = [(2, 1), (3, 2), (17, 1), (136, 1), (136, 136), (64, 64), (200, 5), (250, 3), (300, 2), (421, 1),
           (500, 2), (640, 1), (700, 3), (1024, 2), (1024, 32), (7, 1000)]


def _relations(rng, case):
    """Build side with `dups` copies per key on average, probe side drawn uniformly or by a Zipf law
    from the build keys plus a share of keys that match nothing."""
    inner = int(rng.integers(1, case["inner_max"]))
    outer = int(rng.integers(1, case["outer_max"]))
    distinct = max(1, inner // case["dups"])
    base = np.unique(rng.integers(0 if case["key_zero"] else 1, 2**32, size=distinct, dtype=np.uint64).astype(np.uint32))
    if not case["key_zero"]:
        base = base[base != 0]
        if len(base) == 0:
            base = np.array([7], np.uint32)
    ik = base[rng.integers(0, len(base), size=inner)]
    if case["zipf"] > 0:
        w = 1.0 / np.arange(1, len(base) + 1) ** case["zipf"]
        ok = base[rng.choice(len(base), size=outer, p=w / w.sum())]
    else:
        ok = base[rng.integers(0, len(base), size=outer)]
    miss = rng.random(outer) > case["selectivity"]
    # keys that match nothing: random ones, and small integers (the values a broadcast join picks its empty
    # sentinel from: a probe key equal to the sentinel must not match empty slots)
    strangers = np.where(rng.random(outer) < 0.5, rng.integers(1, 2**32, size=outer, dtype=np.uint64),
                         rng.integers(0 if case["key_zero"] else 1, 64, size=outer, dtype=np.uint64)).astype(np.uint32)
    ok = np.where(miss, strangers, ok)
    iv = rng.integers(0, 2**32, size=inner, dtype=np.uint64).astype(np.uint32)
    ov = rng.integers(0, 2**32, size=outer, dtype=np.uint64).astype(np.uint32)
    return ik, iv, ok.astype(np.uint32), ov


def _cases():
    # HJ_FUZZ_SEED / HJ_FUZZ_CASES: a longer one-off sweep with other draws (the committed default is what CI runs)
    rng = np.random.default_rng(int(os.environ.get("HJ_FUZZ_SEED", "20260101")))
    out = []
    for i in range(int(os.environ.get("HJ_FUZZ_CASES", "36"))):
        f1, f2 = FANOUTS[i % len(FANOUTS)]
        out.append(dict(seed=int(rng.integers(1 << 30)), f1=f1, f2=f2,
                        inner_max=int(rng.choice([300, 40_000, 900_000])),
                        outer_max=int(rng.choice([500, 200_000, 3_000_000])),
                        dups=int(rng.choice([1, 1, 2, 3, 40])), zipf=float(rng.choice([0, 0, 1.1, 2.0])),
                        selectivity=float(rng.choice([1.0, 0.5, 0.0])), key_zero=bool(rng.integers(2)),
                        chunks=int(rng.integers(1, 9)),
                        # every sixth case: the dense final layout (HJGPU_DENSE2), per-chunk pieces in CPRA's join
                        dense2=bool(rng.integers(6) == 0)))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "%dx%d-s%d" % (c["f1"], c["f2"], c["seed"] % 100000))
def test_random_join_matches_numpy(hj, case):
    rng = np.random.default_rng(case["seed"])
    ik, iv, ok, ov = _relations(rng, case)
    want = numpy_join(ik, iv, ok, ov)
    rk, rv, sk, sv = (hj.column(c) for c in (ik, iv, ok, ov))
    if case["dense2"]:
        hj.set_option("dense2", 1)
    try:
        prm = H.PhjParams(fanout1=case["f1"], fanout2=case["f2"], chunks=case["chunks"])
        assert hj.phj(rk, rv, len(ik), sk, sv, len(ok), prm) == want
        assert hj.cpra(rk, rv, len(ik), sk, sv, len(ok), prm) == want
        # the same join with the build side prepared once and the probe side in two batches
        cut = (len(ok) // 3) & ~15
        hj.phj_build(rk, rv, len(ik), max(cut, len(ok) - cut), prm)
        a = hj.phj_probe(sk, sv, cut)
        b = hj.phj_probe(sk.ptr + 4 * cut, sv.ptr + 4 * cut, len(ok) - cut)
        assert tuple((x + y) & ((1 << 64) - 1) for x, y in zip(a, b)) == want
        if not case["key_zero"] and want[0] < 400_000_000:
            assert hj.npj(rk, rv, len(ik), sk, sv, len(ok)) == want
        # rows, for results small enough to sort on the host
        if 0 < want[0] <= 3_000_000:
            block = int(rng.choice([256, 1024, 65536]))
            cap = (want[0] // block + hj.device_info()["compute_units"] * 16 + 8) * block
            jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
            got = hj.phj(rk, rv, len(ik), sk, sv, len(ok), prm, out=(jk, jo, ji, cap, block))
            assert got == want
            rows = sort_rows(jk.download()[:got[0]], jo.download()[:got[0]], ji.download()[:got[0]])
            for a, b in zip(rows, materialised_rows(ik, iv, ok, ov)):
                assert np.array_equal(a, b)
            for c in (jk, jo, ji):
                c.free()
        # the same join by a grouped plan (pass 0 + one planned join per group, phj.cpp:1791-1808): aggregates, and rows
        groups = int(rng.integers(2, 41))
        hj.set_option("group_from", 1)
        hj.set_option("group_always", 1)
        hj.set_option("group_inner", max(1, -(-len(ik) // groups)))
        gprm = H.PhjParams(chunks=case["chunks"])
        assert hj.phj(rk, rv, len(ik), sk, sv, len(ok)) == want
        assert hj.stats()["groups"] >= 2 or len(ik) < 2
        assert hj.cpra(rk, rv, len(ik), sk, sv, len(ok), gprm) == want
        if 0 < want[0] <= 3_000_000:
            block = int(rng.choice([256, 4096]))
            cap = hj.output_capacity(1, len(ok), want[0], block)
            jk, jo, ji = hj.column(cap), hj.column(cap), hj.column(cap)
            got = hj.phj(rk, rv, len(ik), sk, sv, len(ok), None, out=(jk, jo, ji, cap, block))
            assert got == want
            rows = sort_rows(jk.download()[:got[0]], jo.download()[:got[0]], ji.download()[:got[0]])
            for a, b in zip(rows, materialised_rows(ik, iv, ok, ov)):
                assert np.array_equal(a, b)
            for c in (jk, jo, ji):
                c.free()
    finally:
        hj.set_option("dense2", 0)
        hj.set_option("group_always", 0)
        hj.set_option("group_from", 300_000_000)
        hj.set_option("group_inner", 64_000_000)
        for c in (rk, rv, sk, sv):
            c.free()
