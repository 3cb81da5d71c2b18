"""ctypes binding of the CPU oracle (oracle/libhjoracle.so) and, when present,
of the reference's own scalar operators (oracle/_ref/libhjref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


class Result(C.Structure):
    _fields_ = [("count", C.c_uint64), ("sum_keys", C.c_uint64),
                ("sum_outer", C.c_uint64), ("sum_inner", C.c_uint64)]

    def as_tuple(self):
        return (self.count, self.sum_keys, self.sum_outer, self.sum_inner)


class Timing(C.Structure):
    _fields_ = [("seconds", C.c_double), ("seconds_phase", C.c_double * 4)]


class Output(C.Structure):
    _fields_ = [("keys", C.c_void_p), ("outer_vals", C.c_void_p), ("inner_vals", C.c_void_p),
                ("block_size", C.c_size_t), ("block_limit", C.c_size_t),
                ("block_counter", C.POINTER(C.c_size_t))]


def build(force=False):
    # HJ_ORACLE_LIB: another build of the same sources (oracle/Makefile asan / tsan; tests/test_sanitizers.py)
    if os.environ.get("HJ_ORACLE_LIB"):
        return os.environ["HJ_ORACLE_LIB"]
    so = os.path.join(HERE, "libhjoracle.so")
    srcs = [os.path.join(HERE, f) for f in ("hj_oracle.c", "hj_oracle_avx512.c", "hj_oracle.h")]
    if force or not os.path.exists(so) or any(
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so) for src in srcs):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.hjo_hash.restype = C.c_uint32
        L.hjo_hash.argtypes = [C.c_uint32, C.c_uint32]
        L.hjo_simd_available.restype = C.c_int
        L.hjo_set_simd.restype = C.c_int
        L.hjo_set_simd.argtypes = [C.c_int]
        L.hjo_set_unique.restype = C.c_int
        L.hjo_set_unique.argtypes = [C.c_int]
        for f in (L.hjo_thread_beg, L.hjo_thread_end):
            f.restype = C.c_size_t
            f.argtypes = [C.c_size_t] * 4
        L.hjo_odd_prime.restype = C.c_int
        L.hjo_odd_prime.argtypes = [C.c_uint64]
        L.hjo_next_odd_prime.restype = C.c_uint64
        L.hjo_next_odd_prime.argtypes = [C.c_uint64]
        L.hjo_generate.restype = C.c_int
        L.hjo_generate.argtypes = [C.c_size_t, C.c_size_t, C.c_double, C.c_uint32, C.c_uint32,
                                   C.c_uint32, C.c_uint32, _u32p, _u32p, _u32p, _u32p]
        L.hjo_histogram.restype = None
        L.hjo_histogram.argtypes = [_u32p, C.c_size_t, _u32p, C.c_uint32, C.c_size_t]
        L.hjo_partition.restype = None
        L.hjo_partition.argtypes = [_u32p, _u32p, C.c_size_t, _u32p, _u32p, _u32p,
                                    C.c_uint32, C.c_size_t]
        L.hjo_npj_build.restype = None
        L.hjo_npj_build.argtypes = [_u32p, _u32p, C.c_size_t, _u64p, C.c_size_t, C.c_uint32,
                                    C.c_uint32]
        L.hjo_npj_probe.restype = None
        L.hjo_npj_probe.argtypes = [_u32p, _u32p, C.c_size_t, _u64p, C.c_size_t, C.c_uint32,
                                    C.c_uint32, C.POINTER(Result), C.c_void_p, C.c_void_p]
        L.hjo_phj_build.restype = None
        L.hjo_phj_build.argtypes = [_u32p, _u32p, C.c_size_t, _u64p, C.c_size_t,
                                    C.POINTER(C.c_uint32), C.c_uint32]
        L.hjo_phj_probe.restype = None
        L.hjo_phj_probe.argtypes = [_u32p, _u32p, C.c_size_t, _u64p, C.c_size_t,
                                    C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(Result),
                                    C.c_void_p, C.c_void_p]
        L.hjo_close_gaps.restype = C.c_size_t
        L.hjo_close_gaps.argtypes = [_u32p, _u32p, _u32p, C.POINTER(C.c_size_t), C.c_size_t,
                                     C.c_size_t]
        L.hjo_npj.restype = C.c_int
        L.hjo_npj.argtypes = [C.c_int, _u32p, _u32p, C.c_size_t, _u32p, _u32p, C.c_size_t,
                              C.c_double, C.c_uint32, C.POINTER(Result), C.c_void_p,
                              C.POINTER(C.c_size_t), C.POINTER(Timing)]
        L.hjo_phj.restype = C.c_int
        L.hjo_phj.argtypes = [C.c_int, _u32p, _u32p, C.c_size_t, _u32p, _u32p, C.c_size_t,
                              C.c_double, C.c_size_t, C.c_uint32, C.c_uint32,
                              C.POINTER(Result), C.POINTER(Timing)]
        L.hjo_cpra.restype = C.c_int
        L.hjo_cpra.argtypes = [C.c_int, _u32p, _u32p, C.c_size_t, _u32p, _u32p, C.c_size_t,
                               C.c_double, C.c_size_t, C.c_uint32,
                               C.POINTER(Result), C.POINTER(Timing)]
        L.hjo_join_definition.restype = None
        L.hjo_join_definition.argtypes = [_u32p, _u32p, C.c_size_t, _u32p, _u32p, C.c_size_t,
                                          C.POINTER(Result)]
        _lib = L
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


# ---- convenience wrappers --------------------------------------------------

def simd_available():
    """True where the AVX-512 forms of histogram / partition / probe can run on this CPU."""
    return bool(lib().hjo_simd_available())


def set_simd(on):
    """Selects the AVX-512 operator forms (the timed CPU baseline; 2 = with the partition's conflict-serialised
    vector scatter, the reference's shape) or the scalar definitions (default); returns what is in effect
    (0 where the CPU has no AVX-512).  Results are identical either way."""
    return int(lib().hjo_set_simd(int(on)))


def set_unique(on):
    """The reference's -D_UNIQUE build: every probe reports its first match only (default off)."""
    return bool(lib().hjo_set_unique(1 if on else 0))


def join_definition_unique(ik, iv, ok, ov):
    """What a _UNIQUE join must return whatever the table layout: every probe tuple with at least one
    match counts once -> (count, sum_keys, sum_outer); the build payload is one of the key's payloads."""
    ik, ok, ov = _c(ik), _c(ok), _c(ov)
    hit = np.isin(ok, ik)
    return (int(hit.sum()), int(ok[hit].astype(np.uint64).sum()), int(ov[hit].astype(np.uint64).sum()))


def generate(outer, inner, selectivity=1.0, seed=1, unique_factor=0x9E3779B1,
             inner_factor=0x85EBCA6B, outer_factor=0xC2B2AE35):
    """Returns (inner_keys, inner_vals, outer_keys, outer_vals)."""
    ik = np.empty(inner, np.uint32); iv = np.empty(inner, np.uint32)
    ok = np.empty(outer, np.uint32); ov = np.empty(outer, np.uint32)
    rc = lib().hjo_generate(outer, inner, selectivity, seed, unique_factor | 1,
                            inner_factor | 1, outer_factor | 1, ik, iv, ok, ov)
    if rc != 0:
        raise MemoryError("hjo_generate failed")
    return ik, iv, ok, ov


def histogram(keys, factor, partitions):
    counts = np.zeros(partitions, np.uint32)
    lib().hjo_histogram(_c(keys), len(keys), counts, factor, partitions)
    return counts


def partition(keys, vals, factor, partitions):
    keys, vals = _c(keys), _c(vals)
    counts = histogram(keys, factor, partitions)
    ko = np.empty_like(keys); vo = np.empty_like(vals)
    lib().hjo_partition(keys, vals, len(keys), counts, ko, vo, factor, partitions)
    return counts, ko, vo


def join_definition(ik, iv, ok, ov):
    r = Result()
    lib().hjo_join_definition(_c(ik), _c(iv), len(ik), _c(ok), _c(ov), len(ok), C.byref(r))
    return r.as_tuple()


def npj(ik, iv, ok, ov, threads=1, load=0.90, factor=0x9E3779B1, materialize=False,
        block_size=65536, timing=None):
    """npj.cpp run(): returns (count,sum_keys,sum_outer,sum_inner) or, with
    materialize=True, (result, (keys, outer_vals, inner_vals) dense prefix)."""
    r = Result()
    t = timing if timing is not None else Timing()
    ik, iv, ok, ov = _c(ik), _c(iv), _c(ok), _c(ov)
    if not materialize:
        rc = lib().hjo_npj(threads, ik, iv, len(ik), ok, ov, len(ok), load, factor | 1,
                           C.byref(r), None, None, C.byref(t))
        assert rc == 0
        return r.as_tuple()
    # npj.cpp:937-942, 997: block_limit = J_est*1.05/block_size + 2T
    d = max(1, min(len(ik), len(ok)))
    j_est = (len(ok) / d) * (len(ik) / d) * d
    block_limit = int(j_est * 1.05 / block_size) + 2 * threads
    cap = block_limit * block_size
    jk = np.zeros(cap, np.uint32); jo = np.zeros(cap, np.uint32); ji = np.zeros(cap, np.uint32)
    counter = C.c_size_t(0)
    out = Output(jk.ctypes.data, jo.ctypes.data, ji.ctypes.data, block_size, block_limit,
                 C.pointer(counter))
    dense = C.c_size_t(0)
    rc = lib().hjo_npj(threads, ik, iv, len(ik), ok, ov, len(ok), load, factor | 1,
                       C.byref(r), C.byref(out), C.byref(dense), C.byref(t))
    assert rc == 0
    n = dense.value
    return r.as_tuple(), (jk[:n].copy(), jo[:n].copy(), ji[:n].copy())


def phj(ik, iv, ok, ov, threads=1, load=0.4, hash_table_limit=6400,
        thread_factor=0x2545F491, seed=7, timing=None):
    r = Result()
    t = timing if timing is not None else Timing()
    rc = lib().hjo_phj(threads, _c(ik), _c(iv), len(ik), _c(ok), _c(ov), len(ok), load,
                       hash_table_limit, thread_factor | 1, seed, C.byref(r), C.byref(t))
    assert rc == 0
    return r.as_tuple()


def cpra(ik, iv, ok, ov, threads=1, load=0.4, num_partitions=4096, seed=7, timing=None):
    r = Result()
    t = timing if timing is not None else Timing()
    rc = lib().hjo_cpra(threads, _c(ik), _c(iv), len(ik), _c(ok), _c(ov), len(ok), load,
                        num_partitions, seed, C.byref(r), C.byref(t))
    assert rc == 0
    return r.as_tuple()


# ---- the reference's own scalar operators (oracle/_ref) ----------------------

_ref = None


_ref_unique = None


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libhjref.so"))


def ref(unique=False):
    """libhjref.so built by oracle/build_ref.py from /root/reference (unique: the same functions
    compiled with -D_UNIQUE, libhjref_unique.so)."""
    global _ref, _ref_unique
    if unique:
        if _ref_unique is None:
            _ref_unique = _bind_ref(C.CDLL(os.path.join(HERE, "_ref", "libhjref_unique.so")))
        return _ref_unique
    if _ref is None:
        _ref = _bind_ref(C.CDLL(os.path.join(HERE, "_ref", "libhjref.so")))
    return _ref


def _bind_ref(R):
    R.hjref_rand32_init.restype = C.c_void_p
    R.hjref_rand32_init.argtypes = [C.c_uint32]
    R.hjref_rand32_next.restype = C.c_uint32
    R.hjref_rand32_next.argtypes = [C.c_void_p]
    R.hjref_rand32_free.argtypes = [C.c_void_p]
    R.hjref_shuffle.argtypes = [_u32p, C.c_size_t, C.c_void_p]
    R.hjref_unique.argtypes = [_u32p, C.c_size_t, _u32p, C.c_size_t, C.c_uint32,
                               C.c_uint32, C.c_void_p]
    for f in (R.hjref_thread_beg, R.hjref_thread_end):
        f.restype = C.c_size_t
        f.argtypes = [C.c_size_t] * 4
    R.hjref_odd_prime.restype = C.c_int
    R.hjref_odd_prime.argtypes = [C.c_uint64]
    R.hjref_npj_build.argtypes = [_u32p, _u32p, C.c_size_t, _u64p, C.c_size_t,
                                  C.c_uint32, C.c_uint32]
    R.hjref_npj_probe.restype = C.c_size_t
    R.hjref_npj_probe.argtypes = [_u32p, _u32p, C.c_size_t, _u64p, C.c_size_t, C.c_uint32,
                                  C.c_uint32, _u32p, _u32p, _u32p, C.c_size_t, C.c_size_t,
                                  C.POINTER(C.c_size_t)]
    R.hjref_close_gaps.restype = C.c_size_t
    R.hjref_close_gaps.argtypes = [_u32p, _u32p, _u32p, C.POINTER(C.c_size_t), C.c_size_t,
                                   C.c_size_t]
    R.hjref_phj_build.argtypes = [_u32p, _u32p, C.c_size_t, _u64p, C.c_size_t,
                                  C.POINTER(C.c_uint32), C.c_uint32]
    R.hjref_phj_probe.restype = C.c_size_t
    R.hjref_phj_probe.argtypes = [_u32p, _u32p, C.c_size_t, _u64p, C.c_size_t,
                                  C.POINTER(C.c_uint32), C.c_uint32, _u32p, _u32p, _u32p,
                                  C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]
    R.hjref_histogram.argtypes = [_u32p, C.c_size_t, _u32p, C.c_uint32, C.c_size_t]
    R.hjref_partition.argtypes = [_u32p, _u32p, C.c_size_t, _u32p, _u32p, _u32p,
                                  C.c_uint32, C.c_size_t]
    return R
