/*
 * hj_oracle.h — CPU restatement of the reference's hash-join hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The product path is include/hjgpu.h (HIP kernels); it never calls in here.
 *
 * Every function cites the reference file:line (into xtcyclist/
 * hash_join_codes_KNL) whose behaviour it restates.  All data-path arithmetic
 * is 32/64-bit unsigned integer; results are bit-exact by construction.
 *
 * Pinning status: see oracle/README.md ("how this oracle is pinned").
 */
#ifndef HJ_ORACLE_H
#define HJ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Order-free join result (the reference defines no checksum,
 * npj.cpp:905-911 computes join_tuples and never prints it). */
typedef struct {
    uint64_t count;      /* J = number of matching (outer, inner) pairs        */
    uint64_t sum_keys;   /* sum of join_keys[j]        (uint64 wrap-around)    */
    uint64_t sum_outer;  /* sum of join_outer_vals[j]  (probe-side payload)    */
    uint64_t sum_inner;  /* sum of join_inner_vals[j]  (build-side payload)    */
} hjo_result;

/* Optional materialised output, reference block protocol (npj.cpp:244-246,
 * 312-316): three columns of block_limit*block_size slots; a worker claims
 * block b = fetch_add(counter) and fills [b*block_size, (b+1)*block_size).
 * After hjo_close_gaps the prefix [0, count) is dense. Pass NULL to aggregate
 * only. */
typedef struct {
    uint32_t *keys;          /* join_keys        */
    uint32_t *outer_vals;    /* join_outer_vals  */
    uint32_t *inner_vals;    /* join_inner_vals  */
    size_t block_size;       /* power of two; reference: 65536 (npj.cpp:945)  */
    size_t block_limit;
    volatile size_t *block_counter;
} hjo_output;

/* AVX-512 forms of histogram / partition / probe (hj_oracle_avx512.c) for the timed CPU
 * baseline: off by default; hjo_set_simd(1) turns them on where the CPU has AVX-512 and
 * returns what is in effect; hjo_set_simd(2) additionally moves the partition's tuples with the
 * reference's conflict-serialised vector scatter (phj.cpp:1099-1160) instead of scalar stores.
 * Results are identical to the scalar definitions. */
int hjo_simd_available(void);
int hjo_set_simd(int on);
/* The reference's -D_UNIQUE build (npj.cpp:288-290, 436-438; phj.cpp:459, 635-637): a probe tuple
 * reports its FIRST match only.  Off by default; applies to every probe below (operators and whole
 * joins); returns what is in effect.  At one thread "first" is the build tuple inserted first. */
int hjo_set_unique(int on);

/* ---- primitives ------------------------------------------------------- */
/* mulhi32: ((uint64)x * n) >> 32   (npj.cpp:200-201, phj.cpp:83-100, 721-722) */
uint32_t hjo_hash(uint32_t x, uint32_t n);
/* npj.cpp:516-529 */
size_t hjo_thread_beg(size_t size, size_t alignment, size_t thread, size_t threads);
size_t hjo_thread_end(size_t size, size_t alignment, size_t thread, size_t threads);
/* phj.cpp:281-287 */
int hjo_odd_prime(uint64_t x);
/* "for (buckets |= 1 ; !odd_prime(buckets) ; buckets += 2)" phj.cpp:1901 */
uint64_t hjo_next_odd_prime(uint64_t x);

/* MT19937, npj.cpp:133-175 (rand32_init / rand32_next) */
typedef struct { uint32_t num[625]; size_t index; } hjo_rand32;
void hjo_rand32_init(hjo_rand32 *s, uint32_t seed);
uint32_t hjo_rand32_next(hjo_rand32 *s);

/* ---- data generator: intact semantics cpra2.cpp:1578-1696 at T=1 ------- */
/* unique() cpra2.cpp:1544-1570, shuffle() 1530-1542. selectivity in [0,1].
 * Returns 0 on success. Columns must hold `inner` / `outer` uint32 each. */
int hjo_generate(size_t outer, size_t inner, double selectivity, uint32_t seed,
                 uint32_t unique_factor, uint32_t inner_factor, uint32_t outer_factor,
                 uint32_t *inner_keys, uint32_t *inner_vals,
                 uint32_t *outer_keys, uint32_t *outer_vals);
void hjo_shuffle(uint32_t *data, size_t size, hjo_rand32 *gen);
void hjo_unique(uint32_t *keys, size_t size, uint32_t *table, size_t buckets,
                uint32_t factor, uint32_t empty, hjo_rand32 *gen);

/* ---- partitioning: phj.cpp:693-772 / 1029-1231 (scalar forms 1295-1455,
 * cpra2.cpp:730-796) ------------------------------------------------------ */
void hjo_histogram(const uint32_t *keys, size_t size, uint32_t *counts,
                   uint32_t factor, size_t partitions);
/* Output = relation permuted so that partition p occupies
 * [sum_{q<p} counts[q], +counts[p]); order inside a partition = input order
 * (the reference leaves it unspecified). */
void hjo_partition(const uint32_t *keys, const uint32_t *vals, size_t size,
                   const uint32_t *counts, uint32_t *keys_out, uint32_t *vals_out,
                   uint32_t factor, size_t partitions);
/* phj.cpp:1263-1291: offsets[p] for `thread`, aggr_counts[p] totals; returns n. */
size_t hjo_interleave(uint32_t **counts, uint32_t *offsets, uint32_t *aggr_counts,
                      size_t partitions, size_t thread, size_t threads);
/* Scatter one thread's range to precomputed offsets (partition_shared,
 * phj.cpp:877-1028 without the DDR/HBM ratio split). offsets[] is advanced. */
void hjo_partition_shared(const uint32_t *keys, const uint32_t *vals, size_t size,
                          uint32_t *offsets, uint32_t *keys_out, uint32_t *vals_out,
                          uint32_t factor, size_t partitions);

/* ---- NPJ operators: npj.cpp:190-212 (build), 412-445 (probe), 475-514 ---- */
void hjo_npj_build(const uint32_t *keys, const uint32_t *vals, size_t size,
                   volatile uint64_t *table, size_t buckets, uint32_t factor,
                   uint32_t empty);
/* Returns this worker's end offset `o` when out != NULL (npj.cpp:216), else 0.
 * *o_inout carries the cursor between calls (first call: claim a block). */
void hjo_npj_probe(const uint32_t *keys, const uint32_t *vals, size_t size,
                   const uint64_t *table, size_t buckets, uint32_t factor,
                   uint32_t empty, hjo_result *agg, const hjo_output *out,
                   size_t *o_inout);
size_t hjo_close_gaps(uint32_t *keys, uint32_t *vals, uint32_t *tabs,
                      const size_t *offsets, size_t count, size_t block_size);

/* ---- PHJ/CPRA per-partition operators: phj.cpp:577-603, 605-647 --------- */
void hjo_phj_build(const uint32_t *keys, const uint32_t *vals, size_t size,
                   uint64_t *table, size_t buckets, const uint32_t factor[2],
                   uint32_t empty);
void hjo_phj_probe(const uint32_t *keys, const uint32_t *vals, size_t size,
                   const uint64_t *table, size_t buckets, const uint32_t factor[2],
                   uint32_t empty, hjo_result *agg, const hjo_output *out,
                   size_t *o_inout);

/* ---- whole joins (orchestration restated from run / run_hj) -------------- */
typedef struct {
    double seconds;          /* timed region as the reference: table init + build
                                + probe (+ partitioning), data already in memory */
    double seconds_phase[4]; /* npj: init+build, probe, close_gaps; phj/cpra:
                                pass-1, local passes, join, - */
} hjo_timing;

/* npj.cpp:769-927 with `threads` pthreads; load = hash_table_load (0.90 in
 * the reference, npj.cpp:944); factor = odd hash multiplier. */
int hjo_npj(int threads,
            const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
            const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
            double load, uint32_t factor,
            hjo_result *res, const hjo_output *out, size_t *join_tuples_dense,
            hjo_timing *timing);

/* phj.cpp:1646-1949 including the commented-out join loop 1869-1924.
 * thread_factor = pass-1 (cross-thread) factor (phj.cpp:2167), seed = MT seed
 * for the later pass / table factors (phj.cpp:1823, 1873-1876).
 * hash_table_limit = 6400, load = 0.4 in the reference (phj.cpp:1976-1977). */
int hjo_phj(int threads,
            const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
            const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
            double load, size_t hash_table_limit, uint32_t thread_factor, uint32_t seed,
            hjo_result *res, hjo_timing *timing);

/* cpra2.cpp:1697-1986: every thread partitions its own chunk into
 * num_partitions (4096 in the reference, cpra2.cpp:2023), owners gather. */
int hjo_cpra(int threads,
             const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
             const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
             double load, size_t num_partitions, uint32_t seed,
             hjo_result *res, hjo_timing *timing);

/* Brute-force definition of the join result (sort-free, hash-free for small
 * inputs: O(inner*outer) when inner*outer <= 2^26, otherwise sort-merge).
 * Independent cross-check of everything above. */
void hjo_join_definition(const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
                         const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
                         hjo_result *res);

#ifdef __cplusplus
}
#endif
#endif
