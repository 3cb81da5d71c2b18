"""Shared helpers for the parity tests (numpy only)."""
import numpy as np


def mulhi_hash(keys, factor, n):
    """H(key, f, N) = ((uint32)(key*f) * N) >> 32   (npj.cpp:200-201)."""
    x = (keys.astype(np.uint64) * np.uint64(factor)) & np.uint64(0xFFFFFFFF)
    return ((x * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def pairs(keys, vals):
    return (keys.astype(np.uint64) << np.uint64(32)) | vals.astype(np.uint64)


def numpy_join(ik, iv, ok, ov):
    """Independent definition of the result: (count, sum_keys, sum_outer, sum_inner)."""
    order = np.argsort(ik, kind="stable")
    sk, sv = ik[order], iv[order].astype(np.uint64)
    csum = np.concatenate([np.zeros(1, np.uint64), np.cumsum(sv, dtype=np.uint64)])   # stays uint64
    lo = np.searchsorted(sk, ok, side="left")
    hi = np.searchsorted(sk, ok, side="right")
    mult = (hi - lo).astype(np.uint64)
    mask = np.uint64(0xFFFFFFFFFFFFFFFF)
    count = int(mult.sum())
    sum_keys = int((ok.astype(np.uint64) * mult).sum(dtype=np.uint64)) & int(mask)
    sum_outer = int((ov.astype(np.uint64) * mult).sum(dtype=np.uint64)) & int(mask)
    sum_inner = int((csum[hi] - csum[lo]).sum(dtype=np.uint64)) & int(mask)
    return (count, sum_keys, sum_outer, sum_inner)


def materialised_rows(ik, iv, ok, ov):
    """All result rows (key, outer_val, inner_val) sorted lexicographically."""
    order = np.argsort(ik, kind="stable")
    sk, sv = ik[order], iv[order]
    lo = np.searchsorted(sk, ok, side="left")
    hi = np.searchsorted(sk, ok, side="right")
    mult = hi - lo
    rows_o = np.repeat(np.arange(len(ok)), mult)
    starts = np.repeat(lo, mult)
    within = np.arange(len(rows_o)) - np.repeat(np.cumsum(mult) - mult, mult)
    rk = ok[rows_o]; ro = ov[rows_o]; ri = sv[starts + within]
    idx = np.lexsort((ri, ro, rk))
    return rk[idx], ro[idx], ri[idx]


def sort_rows(k, o, i):
    idx = np.lexsort((i, o, k))
    return k[idx], o[idx], i[idx]


def make_relations(oracle, outer, inner, seed=1, selectivity=1.0):
    return oracle.generate(outer, inner, selectivity=selectivity, seed=seed)
