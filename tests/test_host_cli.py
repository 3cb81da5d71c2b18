"""CPU tests of the C++ host programs: ./write keeps the reference's file names
and raw uint32 format and equals the oracle's restatement of the reference
generator; the join programs fail loudly without input files / without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import hash_join_codes_knl_amd as H

LIB = os.path.join(os.path.dirname(os.path.abspath(H.__file__)), "lib")


@pytest.fixture(scope="module")
def programs():
    H.build.build_all(verbose=False)
    return {n: os.path.join(LIB, n) for n in ("npj", "phj", "cpra", "write")}


def _factors(oracle, seed):
    class RS(C.Structure):
        _fields_ = [("num", C.c_uint32 * 625), ("index", C.c_size_t)]
    s = RS()
    L = oracle.lib()
    L.hjo_rand32_init(C.byref(s), seed ^ 0x5bd1e995)
    L.hjo_rand32_next.restype = C.c_uint32
    return [L.hjo_rand32_next(C.byref(s)) | 1 for _ in range(3)]


@pytest.mark.parametrize("outer,inner,sel", [(5000, 1200, 1.0), (700, 4000, 1.0), (3000, 3000, 0.25)])
def test_write_matches_reference_generator(programs, oracle, tmp_path, outer, inner, sel):
    env = dict(os.environ, HJ_SEED="7")
    subprocess.check_call([programs["write"], "4", str(outer), str(inner), str(sel)], cwd=tmp_path, env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cols = {}
    for prefix, n in (("ik", inner), ("iv", inner), ("ok", outer), ("ov", outer)):
        path = tmp_path / ("%s_%d.txt" % (prefix, n))            # write.cpp:1824-1865 naming
        assert path.stat().st_size == 4 * n
        cols[prefix] = np.fromfile(path, dtype="<u4")
    uf, fi, fo = _factors(oracle, 7)
    ik, iv, ok, ov = oracle.generate(outer, inner, selectivity=sel, seed=7, unique_factor=uf,
                                     inner_factor=fi, outer_factor=fo)
    assert np.array_equal(cols["ik"], ik) and np.array_equal(cols["iv"], iv)
    assert np.array_equal(cols["ok"], ok) and np.array_equal(cols["ov"], ov)
    assert (cols["ik"] != 0).all()


def test_write_zipf_skews_probe_side(programs, tmp_path):
    subprocess.check_call([programs["write"], "1", "200000", "1000", "1.0", "1.0"], cwd=tmp_path,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ok = np.fromfile(tmp_path / "ok_200000.txt", dtype="<u4")
    ik = np.fromfile(tmp_path / "ik_1000.txt", dtype="<u4")
    assert np.isin(ok, ik).all() and len(np.unique(ok)) == 1000
    counts = np.sort(np.unique(ok, return_counts=True)[1])[::-1]
    assert counts[0] > 20 * counts[500]          # heavy head, unlike the uniform picks


@pytest.mark.parametrize("prog", ["npj", "phj", "cpra"])
def test_join_programs_report_missing_input(programs, tmp_path, prog):
    p = subprocess.run([programs[prog], "8", "100", "100"], cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode == 2 and "cannot open" in p.stderr and p.stdout == ""
