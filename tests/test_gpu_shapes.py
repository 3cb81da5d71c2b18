"""GPU tests of workload shapes away from the benchmark's: tiny build side against
a large probe side (slices), build side larger than probe side, heavy skew
(multi-fill overflow + chained fallback), selectivity 0, many duplicates on both sides."""
import numpy as np
import pytest

import hash_join_codes_knl_amd as H
from helpers import numpy_join

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
