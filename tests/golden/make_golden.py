#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small input relations + the outputs that the
REFERENCE'S OWN scalar operator code (oracle/_ref/libhjref.so, compiled from
/root/reference by oracle/build_ref.py) produces for them.

Run in the authoring container only (needs /root/reference):
    python oracle/build_ref.py && python tests/golden/make_golden.py
The fixtures are data (inputs and expected outputs); no reference source text
is stored.  Reference functions exercised, per fixture:
  rand32_init/next, unique, shuffle (npj.cpp:133-175, 560-600)  -> the relations themselves
  histogram_s / partition_s (cpra2.cpp:730-796)                  -> hist_*, part_*
  build / probe / close_gaps (npj.cpp:190-212, 412-445, 475-514) -> npj_*
  build_s / probe_s (phj.cpp:577-647)                            -> phj_*
  the same probe / probe_s compiled with -D_UNIQUE (libhjref_unique.so: npj.cpp:436-438,
  phj.cpp:635-637)                                               -> npj_result_unique, phj_result_unique
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
HIST_CASES = [(0x9E3779B1, 7), (0x85EBCA6B, 64), (0x9E3779B1, 1000)]
PART_CASES = [(0x9E3779B1, 7), (0x85EBCA6B, 64)]
NPJ_FACTOR, NPJ_LOAD = 0x9E3779B1, 0.90           # npj.cpp:944
PHJ_FACTORS, PHJ_LOAD = (0xC2B2AE35, 0x27D4EB2F), 0.4   # phj.cpp:1976; (f0-f1)&3 != 0


def aligned(n, dtype=np.uint32):
    b = np.zeros(n + 32, dtype)
    o = (-b.ctypes.data % 64) // b.itemsize
    return b[o:o + n]


def ref_generate(R, outer, inner, selectivity, seed, unique_factor, inner_factor, outer_factor):
    """generate_data_for_join (cpra2.cpp:1578-1696) at T = 1, driven through the
    reference's own rand32 / unique / shuffle."""
    d = min(inner, outer)
    join_d = int(d * selectivity)
    distinct = 2 * d - join_d
    buckets = distinct * 2 + 1
    while not R.hjref_odd_prime(buckets):
        buckets += 2
    gen = R.hjref_rand32_init(seed)
    uniq = np.zeros(distinct, np.uint32)
    table = np.zeros(buckets, np.uint32)
    R.hjref_unique(uniq, distinct, table, buckets, unique_factor, 0, gen)
    ik = np.zeros(inner, np.uint32)
    u = 0
    for i in range(inner):
        if u != d:
            ik[i] = uniq[u]; u += 1
        else:
            ik[i] = uniq[(R.hjref_rand32_next(gen) * d) >> 32]
    ou = uniq[d - join_d:]
    ok = np.zeros(outer, np.uint32)
    u = 0
    for o in range(outer):
        if u != d:
            ok[o] = ou[u]; u += 1
        else:
            ok[o] = ou[(R.hjref_rand32_next(gen) * d) >> 32]
    R.hjref_shuffle(ik, inner, gen)
    R.hjref_shuffle(ok, outer, gen)
    R.hjref_rand32_free(gen)
    iv = (ik.astype(np.uint64) * inner_factor).astype(np.uint32)
    ov = (ok.astype(np.uint64) * outer_factor).astype(np.uint32)
    return ik, iv, ok, ov


def part_sums(keys, vals, counts):
    off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
    sk = np.add.reduceat(np.concatenate([keys.astype(np.uint64), [0]]), np.minimum(off[:-1], len(keys)))
    sv = np.add.reduceat(np.concatenate([vals.astype(np.uint64), [0]]), np.minimum(off[:-1], len(vals)))
    sk = np.where(counts == 0, 0, sk).astype(np.uint64)
    sv = np.where(counts == 0, 0, sv).astype(np.uint64)
    return sk, sv


def expected_from_reference(R, ik, iv, ok, ov, allow_npj=True, U=None):
    """U: the -D_UNIQUE build of the same reference functions (first match only)."""
    out = {}
    # ---- histogram_s / partition_s ----
    for idx, (f, F) in enumerate(HIST_CASES):
        c = np.zeros(F, np.uint32)
        R.hjref_histogram(ok, len(ok), c, f, F)
        out["hist_%d" % idx] = c
    for idx, (f, F) in enumerate(PART_CASES):
        c = np.zeros(F, np.uint32)
        R.hjref_histogram(ok, len(ok), c, f, F)
        ko, vo = aligned(len(ok)), aligned(len(ok))
        R.hjref_partition(ok, ov, len(ok), c, ko, vo, f, F)
        sk, sv = part_sums(ko, vo, c)
        out["part_%d_counts" % idx] = c
        out["part_%d_sum_keys" % idx] = sk
        out["part_%d_sum_vals" % idx] = sv
        if idx == 0:
            out["part_0_keys"] = ko.copy(); out["part_0_vals"] = vo.copy()
    block = 1024
    cap = (int(len(ok) * max(1, len(ik) // max(1, min(len(ik), len(ok)))) * 1.05) // block + 4) * block
    # ---- NPJ: set + build + probe + close_gaps ----
    if allow_npj:
        buckets = int(len(ik) / NPJ_LOAD)
        table = np.zeros(buckets, np.uint64)
        R.hjref_npj_build(ik, iv, len(ik), table, buckets, NPJ_FACTOR, 0)
        jk, jo, ji = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        counter = C.c_size_t(0)
        end = R.hjref_npj_probe(ok, ov, len(ok), table, buckets, NPJ_FACTOR, 0, jk, jo, ji,
                                block, cap // block, C.byref(counter))
        offs = (C.c_size_t * 1)(end)
        n = R.hjref_close_gaps(jk, jo, ji, offs, 1, block)
        out["npj_table_sorted"] = np.sort(table)
        out["npj_result"] = np.array([n, jk[:n].astype(np.uint64).sum(), jo[:n].astype(np.uint64).sum(),
                                      ji[:n].astype(np.uint64).sum()], np.uint64)
        if U is not None:
            jk[:] = 0; jo[:] = 0; ji[:] = 0
            counter = C.c_size_t(0)
            end = U.hjref_npj_probe(ok, ov, len(ok), table, buckets, NPJ_FACTOR, 0, jk, jo, ji,
                                    block, cap // block, C.byref(counter))
            offs = (C.c_size_t * 1)(end)
            n = U.hjref_close_gaps(jk, jo, ji, offs, 1, block)
            out["npj_result_unique"] = np.array([n, jk[:n].astype(np.uint64).sum(), jo[:n].astype(np.uint64).sum(),
                                                 ji[:n].astype(np.uint64).sum()], np.uint64)
    # ---- PHJ operators on the whole relation as one partition ----
    buckets = int(O.lib().hjo_next_odd_prime(int(len(ik) / PHJ_LOAD)))
    table = np.zeros(buckets, np.uint64)
    fac = (C.c_uint32 * 2)(*PHJ_FACTORS)
    empty = 0
    if (ik == 0).any():
        empty = 1
        while (ik == empty).any() or (ok == empty).any():
            empty += 1
    R.hjref_phj_build(ik, iv, len(ik), table, buckets, fac, empty)
    jk, jo, ji = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    counter = C.c_size_t(1)          # block 0 is the initial `offset`
    end = R.hjref_phj_probe(ok, ov, len(ok), table, buckets, fac, empty, jk, jo, ji, 0,
                            block, cap // block, C.byref(counter))
    offs = (C.c_size_t * 1)(end)
    n = R.hjref_close_gaps(jk, jo, ji, offs, 1, block)
    out["phj_buckets"] = np.array([buckets, empty], np.uint64)
    out["phj_table_sorted"] = np.sort(table)
    out["phj_result"] = np.array([n, jk[:n].astype(np.uint64).sum(), jo[:n].astype(np.uint64).sum(),
                                  ji[:n].astype(np.uint64).sum()], np.uint64)
    if U is not None:
        jk[:] = 0; jo[:] = 0; ji[:] = 0
        counter = C.c_size_t(1)
        end = U.hjref_phj_probe(ok, ov, len(ok), table, buckets, fac, empty, jk, jo, ji, 0,
                                block, cap // block, C.byref(counter))
        offs = (C.c_size_t * 1)(end)
        n = U.hjref_close_gaps(jk, jo, ji, offs, 1, block)
        out["phj_result_unique"] = np.array([n, jk[:n].astype(np.uint64).sum(), jo[:n].astype(np.uint64).sum(),
                                             ji[:n].astype(np.uint64).sum()], np.uint64)
    return out


def main():
    if not O.ref_available():
        sys.exit("oracle/_ref/libhjref.so missing: run oracle/build_ref.py where /root/reference exists")
    R = O.ref()
    U = O.ref(unique=True)
    gen = R.hjref_rand32_init(5489)
    stream = np.array([R.hjref_rand32_next(gen) for _ in range(1000)], np.uint32)
    R.hjref_rand32_free(gen)
    np.savez_compressed(os.path.join(HERE, "rand32_seed5489.npz"), stream=stream)

    fixtures = {
        # name: (outer, inner, selectivity, seed)
        "unique_2k_16k": (16384, 2048, 1.0, 11),
        "dups16_8k_512": (512, 8192, 1.0, 12),          # config-1 shape: 16 copies per build key
        "sel_half_4k_8k": (8192, 4096, 0.5, 13),
    }
    for name, (outer, inner, sel, seed) in fixtures.items():
        uf, fi, fo = 0x9E3779B1, 0x2545F491, 0x85EBCA6B
        ik, iv, ok, ov = ref_generate(R, outer, inner, sel, seed, uf, fi, fo)
        exp = expected_from_reference(R, ik, iv, ok, ov, U=U)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), inner_keys=ik, inner_vals=iv,
                            outer_keys=ok, outer_vals=ov,
                            gen_params=np.array([outer, inner, int(sel * 1000), seed, uf, fi, fo], np.uint64),
                            **exp)
        print(name, "npj", exp["npj_result"], "phj", exp["phj_result"], "unique", exp["npj_result_unique"], exp["phj_result_unique"])
    # _UNIQUE with build duplicates that carry DIFFERENT payloads: which duplicate a probe reports is then
    # visible in sum_inner_vals.  At one thread the reference's tables hold a key's duplicates in insertion
    # order along the probe sequence, so "first match" = the build tuple that comes first in the relation.
    ik, iv, ok, ov = ref_generate(R, 3000, 12000, 0.75, 14, 0x9E3779B1, 0x2545F491, 0x85EBCA6B)
    iv = ((np.arange(len(ik), dtype=np.uint64) * 0x9E3779B97F4A7C15 >> np.uint64(29)) & 0xFFFFFFFF).astype(np.uint32)
    exp = expected_from_reference(R, ik, iv, ok, ov, U=U)
    np.savez_compressed(os.path.join(HERE, "dups4_distinct_payloads.npz"), inner_keys=ik, inner_vals=iv,
                        outer_keys=ok, outer_vals=ov, **exp)
    print("dups4_distinct_payloads npj", exp["npj_result"], "unique", exp["npj_result_unique"], exp["phj_result_unique"])
    # sentinel edge case: key 0 and extreme keys present (PHJ/CPRA only; NPJ reserves key 0)
    rng = np.random.default_rng(99)
    ik = np.unique(rng.integers(1, 2**32, size=1500, dtype=np.uint64).astype(np.uint32))[:1000].copy()
    ik[0], ik[1], ik[2] = 0, 0xFFFFFFFF, 1
    ik = np.unique(ik)
    ok = ik[rng.integers(0, len(ik), size=6000)]
    ok[:4] = [0, 0, 0xFFFFFFFF, 1]
    iv = (ik.astype(np.uint64) * 0x2545F491 + 7).astype(np.uint32)
    ov = (ok.astype(np.uint64) * 0x85EBCA6B + 3).astype(np.uint32)
    exp = expected_from_reference(R, ik, iv, ok, ov, allow_npj=False, U=U)
    np.savez_compressed(os.path.join(HERE, "key_zero_and_extremes.npz"), inner_keys=ik, inner_vals=iv,
                        outer_keys=ok, outer_vals=ov, **exp)
    print("key_zero_and_extremes phj", exp["phj_result"], exp["phj_buckets"])


if __name__ == "__main__":
    main()
