/*
 * hj_oracle.c — CPU restatement of the reference hash-join hot path.
 * TEST INFRASTRUCTURE ONLY (see hj_oracle.h). Plain C11 + pthreads.
 *
 * Citations are file:line into the reference (xtcyclist/hash_join_codes_KNL).
 * The restatement follows the reference's *scalar* definitions of each
 * operator (the AVX-512 loops compute the same function 16 lanes at a time)
 * and its run()/run_hj() phase structure.
 */
#define _GNU_SOURCE
#include "hj_oracle.h"

#include <assert.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* primitives                                                               */
/* ------------------------------------------------------------------------ */

static double now_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* npj.cpp:200-201 / phj.cpp:721-722: h = ((uint64)(uint32)(key*factor) * N) >> 32.
 * Callers pass x = key*factor already reduced mod 2^32. */
uint32_t hjo_hash(uint32_t x, uint32_t n)
{
    return (uint32_t)(((uint64_t)x * (uint64_t)n) >> 32);
}

/* Wide form used where N is a size_t bucket count (npj.cpp:200-201). */
static inline size_t hash_wide(uint32_t x, size_t n)
{
    return (size_t)(((unsigned __int128)x * (unsigned __int128)n) >> 32);
}

/* npj.cpp:516-521 */
size_t hjo_thread_beg(size_t size, size_t alignment, size_t thread, size_t threads)
{
    size_t part = (size / threads) & ~(alignment - 1);
    return part * thread;
}

/* npj.cpp:523-529 */
size_t hjo_thread_end(size_t size, size_t alignment, size_t thread, size_t threads)
{
    size_t part = (size / threads) & ~(alignment - 1);
    return (thread + 1 == threads) ? size : part * (thread + 1);
}

/* phj.cpp:281-287 (trial division by odd d; caller guarantees x odd) */
int hjo_odd_prime(uint64_t x)
{
    for (uint64_t d = 3; d * d <= x; d += 2)
        if (x % d == 0) return 0;
    return 1;
}

/* phj.cpp:1901 */
uint64_t hjo_next_odd_prime(uint64_t x)
{
    x |= 1;
    while (!hjo_odd_prime(x)) x += 2;
    return x;
}

/* MT19937: npj.cpp:138-148 (seeding without the "+ i" term of the textbook
 * generator — the reference's variant is what is restated) */
void hjo_rand32_init(hjo_rand32 *s, uint32_t seed)
{
    s->num[0] = seed;
    for (size_t i = 0; i != 623; ++i)
        s->num[i + 1] = 0x6c078965u * (s->num[i] ^ (s->num[i] >> 30));
    s->index = 624;
}

/* npj.cpp:149-175 */
uint32_t hjo_rand32_next(hjo_rand32 *s)
{
    uint32_t *n = s->num;
    if (s->index == 624) {
        size_t i = 0;
        uint32_t y;
        for (; i != 227; ++i) {
            y = (n[i] & 0x80000000u) + (n[i + 1] & 0x7fffffffu);
            n[i] = n[i + 397] ^ (y >> 1) ^ (0x9908b0dfu & (0u - (y & 1u)));
        }
        n[624] = n[0];
        for (; i != 624; ++i) {
            y = (n[i] & 0x80000000u) + (n[i + 1] & 0x7fffffffu);
            n[i] = n[i - 227] ^ (y >> 1) ^ (0x9908b0dfu & (0u - (y & 1u)));
        }
        s->index = 0;
    }
    uint32_t y = n[s->index++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* ------------------------------------------------------------------------ */
/* generator (cpra2.cpp:1578-1696 at T = 1)                                  */
/* ------------------------------------------------------------------------ */

/* cpra2.cpp:1530-1542: forward Fisher-Yates, j = ((rand32*(n-i))>>32)+i */
void hjo_shuffle(uint32_t *data, size_t size, hjo_rand32 *gen)
{
    for (size_t i = 0; i != size; ++i) {
        uint64_t j = hjo_rand32_next(gen);
        j = ((j * (uint64_t)(size - i)) >> 32) + i;
        uint32_t t = data[i];
        data[i] = data[j];
        data[j] = t;
    }
}

/* cpra2.cpp:1544-1570: rejection-sample distinct non-`empty` keys through a
 * linear-probing set (single-threaded here, so the CAS is a plain store). */
void hjo_unique(uint32_t *keys, size_t size, uint32_t *table, size_t buckets,
                uint32_t factor, uint32_t empty, hjo_rand32 *gen)
{
    size_t i = 0;
    while (i != size) {
        uint32_t key;
        do key = hjo_rand32_next(gen); while (key == empty);
        size_t h = hash_wide(key * factor, buckets);
        for (;;) {
            uint32_t tab = table[h];
            if (tab == key) break;              /* duplicate draw: discard */
            if (tab == empty) {
                table[h] = key;
                keys[i++] = key;
                break;
            }
            if (++h == buckets) h = 0;
        }
    }
}

int hjo_generate(size_t outer, size_t inner, double selectivity, uint32_t seed,
                 uint32_t unique_factor, uint32_t inner_factor, uint32_t outer_factor,
                 uint32_t *inner_keys, uint32_t *inner_vals,
                 uint32_t *outer_keys, uint32_t *outer_vals)
{
    /* write.cpp:1687-1689 / cpra2.cpp:2024-2026 */
    size_t outer_distinct = inner < outer ? inner : outer;
    size_t inner_distinct = outer_distinct;
    size_t join_distinct = (size_t)((double)outer_distinct * selectivity);
    if (join_distinct > outer_distinct) join_distinct = outer_distinct;
    size_t distinct = outer_distinct + inner_distinct - join_distinct;
    if (distinct == 0) return 0;
    /* cpra2.cpp:2087-2089 */
    size_t buckets = distinct * 2 + 1;
    while (!hjo_odd_prime(buckets)) buckets += 2;
    uint32_t *uniq = (uint32_t *)malloc(distinct * sizeof(uint32_t));
    uint32_t *table = (uint32_t *)calloc(buckets, sizeof(uint32_t));
    if (!uniq || !table) { free(uniq); free(table); return -1; }
    hjo_rand32 gen;
    hjo_rand32_init(&gen, seed);
    /* cpra2.cpp:1603-1606 */
    hjo_unique(uniq, distinct, table, buckets, unique_factor, 0, &gen);
    free(table);
    /* cpra2.cpp:1610-1628: each distinct key once, then random repeats */
    size_t u = 0;
    for (size_t i = 0; i != inner; ++i) {
        if (u != inner_distinct) inner_keys[i] = uniq[u++];
        else {
            uint64_t r = hjo_rand32_next(&gen);
            inner_keys[i] = uniq[(r * inner_distinct) >> 32];
        }
    }
    /* cpra2.cpp:1631-1650 */
    const uint32_t *outer_unique = &uniq[inner_distinct - join_distinct];
    u = 0;
    for (size_t o = 0; o != outer; ++o) {
        if (u != outer_distinct) outer_keys[o] = outer_unique[u++];
        else {
            uint64_t r = hjo_rand32_next(&gen);
            outer_keys[o] = outer_unique[(r * outer_distinct) >> 32];
        }
    }
    free(uniq);
    /* cpra2.cpp:1654-1660 */
    hjo_shuffle(inner_keys, inner, &gen);
    hjo_shuffle(outer_keys, outer, &gen);
    /* cpra2.cpp:1663-1674 */
    for (size_t i = 0; i != inner; ++i) inner_vals[i] = inner_keys[i] * inner_factor;
    for (size_t o = 0; o != outer; ++o) outer_vals[o] = outer_keys[o] * outer_factor;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* partitioning                                                             */
/* ------------------------------------------------------------------------ */

/* AVX-512 forms of the three hot operators (hj_oracle_avx512.c), off unless hjo_set_simd(1):
 * the scalar definitions below stay the oracle; the vector forms are the timed CPU baseline. */
int hjo_avx512_compiled(void);
void hjo_histogram_avx512(const uint32_t *, size_t, uint32_t *, uint32_t, size_t);
void hjo_partition_avx512(const uint32_t *, const uint32_t *, size_t, const uint32_t *, uint32_t *,
                          uint32_t *, uint32_t, size_t);
void hjo_partition_shared_avx512(const uint32_t *, const uint32_t *, size_t, uint32_t *, uint32_t *,
                                 uint32_t *, uint32_t, size_t);
void hjo_phj_probe_avx512(const uint32_t *, const uint32_t *, size_t, const uint64_t *, size_t,
                          const uint32_t[2], uint32_t, hjo_result *);
static int g_simd = 0;

int hjo_simd_available(void)
{
#if defined(__x86_64__)
    return hjo_avx512_compiled() && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512cd") &&
           __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
#else
    return 0;
#endif
}

void hjo_avx512_vector_scatter(int on);
int hjo_set_simd(int on)
{
    g_simd = (on && hjo_simd_available()) ? 1 : 0;
    hjo_avx512_vector_scatter(g_simd && on == 2);         /* 2: the partition's conflict-serialised vector scatter */
    return g_simd ? on : 0;
}

/* The reference's -D_UNIQUE build (npj.cpp:288-290, 436-438; phj.cpp:459, 635-637; cpra2.cpp:456, 625, 698):
 * a compile-time switch there, a run-time one here.  The walk of a probe tuple ends at its first match. */
static int g_unique = 0;
int hjo_set_unique(int on)
{
    g_unique = on ? 1 : 0;
    return g_unique;
}

/* phj.cpp:1295-1306 (scalar), vector form 693-772 */
void hjo_histogram(const uint32_t *keys, size_t size, uint32_t *counts,
                   uint32_t factor, size_t partitions)
{
    memset(counts, 0, partitions * sizeof(uint32_t));
#if defined(__x86_64__)
    if (g_simd && partitions < (1u << 27)) { hjo_histogram_avx512(keys, size, counts, factor, partitions); return; }
#endif
    for (size_t i = 0; i != size; ++i)
        counts[hash_wide(keys[i] * factor, partitions)]++;
}

/* phj.cpp:1029-1231: the write-combining buffers only change *when* bytes
 * reach memory, not where; the resulting layout is a counting sort. */
void hjo_partition(const uint32_t *keys, const uint32_t *vals, size_t size,
                   const uint32_t *counts, uint32_t *keys_out, uint32_t *vals_out,
                   uint32_t factor, size_t partitions)
{
#if defined(__x86_64__)
    if (g_simd && partitions < (1u << 27)) {
        hjo_partition_avx512(keys, vals, size, counts, keys_out, vals_out, factor, partitions);
        return;
    }
#endif
    size_t *offsets = (size_t *)malloc((partitions ? partitions : 1) * sizeof(size_t));
    size_t acc = 0;
    for (size_t p = 0; p != partitions; ++p) { offsets[p] = acc; acc += counts[p]; }
    assert(acc == size);
    for (size_t i = 0; i != size; ++i) {
        uint32_t key = keys[i];
        size_t o = offsets[hash_wide(key * factor, partitions)]++;
        keys_out[o] = key;
        vals_out[o] = vals[i];
    }
    free(offsets);
}

/* phj.cpp:1263-1291 */
size_t hjo_interleave(uint32_t **counts, uint32_t *offsets, uint32_t *aggr_counts,
                      size_t partitions, size_t thread, size_t threads)
{
    size_t total = 0;
    for (size_t p = 0; p != partitions; ++p) {
        uint32_t before = 0, all = 0;
        for (size_t t = 0; t != threads; ++t) {
            if (t < thread) before += counts[t][p];
            all += counts[t][p];
        }
        offsets[p] = (uint32_t)(total + before);
        aggr_counts[p] = all;
        total += all;
    }
    return total;
}

/* phj.cpp:877-1028, plain H(key*f, T) meaning (scalar fallback 1412-1417) */
void hjo_partition_shared(const uint32_t *keys, const uint32_t *vals, size_t size,
                          uint32_t *offsets, uint32_t *keys_out, uint32_t *vals_out,
                          uint32_t factor, size_t partitions)
{
#if defined(__x86_64__)
    if (g_simd && partitions < (1u << 27)) {
        hjo_partition_shared_avx512(keys, vals, size, offsets, keys_out, vals_out, factor, partitions);
        return;
    }
#endif
    for (size_t i = 0; i != size; ++i) {
        uint32_t key = keys[i];
        size_t o = offsets[hash_wide(key * factor, partitions)]++;
        keys_out[o] = key;
        vals_out[o] = vals[i];
    }
}

/* ------------------------------------------------------------------------ */
/* output sink shared by the probe operators                                 */
/* ------------------------------------------------------------------------ */

static inline void emit(hjo_result *agg, const hjo_output *out, size_t *o,
                        uint32_t key, uint32_t outer_val, uint32_t inner_val)
{
    agg->count++;
    agg->sum_keys += key;
    agg->sum_outer += outer_val;
    agg->sum_inner += inner_val;
    if (out) {
        /* npj.cpp:426-436 */
        size_t pos = *o;
        out->inner_vals[pos] = inner_val;
        out->outer_vals[pos] = outer_val;
        out->keys[pos] = key;
        if ((++pos & (out->block_size - 1)) == 0) {
            pos = __sync_fetch_and_add(out->block_counter, 1);
            assert(pos <= out->block_limit);
            pos *= out->block_size;
        }
        *o = pos;
    }
}

static inline void claim_first_block(const hjo_output *out, size_t *o)
{
    if (out) {
        /* npj.cpp:418-420 */
        size_t b = __sync_fetch_and_add(out->block_counter, 1);
        assert(b <= out->block_limit);
        *o = b * out->block_size;
    }
}

/* ------------------------------------------------------------------------ */
/* NPJ operators                                                            */
/* ------------------------------------------------------------------------ */

/* npj.cpp:190-212: bucket = (val << 32) | key, linear probing, CAS insert */
void hjo_npj_build(const uint32_t *keys, const uint32_t *vals, size_t size,
                   volatile uint64_t *table, size_t buckets, uint32_t factor,
                   uint32_t empty)
{
    for (size_t i = 0; i != size; ++i) {
        uint32_t key = keys[i];
        uint64_t pair = ((uint64_t)vals[i] << 32) | key;
        size_t h = hash_wide(key * factor, buckets);
        for (;;) {
            /* (the reference reads the bucket through a volatile pointer, npj.cpp:202; a relaxed atomic load is the same
             * instruction and says so to ThreadSanitizer: other workers CAS into the table meanwhile) */
            uint64_t tab = __atomic_load_n(&table[h], __ATOMIC_RELAXED);
            if ((uint32_t)tab == empty &&
                __sync_bool_compare_and_swap(&table[h], tab, pair))
                break;
            if (++h == buckets) h = 0;
        }
    }
}

/* npj.cpp:412-445 (scalar form of 216-364): walk to the first empty bucket,
 * every key match is reported; under hjo_set_unique(1) the walk ends at the first
 * match (#ifdef _UNIQUE break, npj.cpp:436-438). */
void hjo_npj_probe(const uint32_t *keys, const uint32_t *vals, size_t size,
                   const uint64_t *table, size_t buckets, uint32_t factor,
                   uint32_t empty, hjo_result *agg, const hjo_output *out,
                   size_t *o_inout)
{
    size_t o = o_inout ? *o_inout : 0;
    for (size_t i = 0; i != size; ++i) {
        uint32_t key = keys[i];
        uint32_t val = vals[i];
        size_t h = hash_wide(key * factor, buckets);
        uint64_t tab = table[h];
        while ((uint32_t)tab != empty) {
            if ((uint32_t)tab == key) {
                emit(agg, out, &o, key, val, (uint32_t)(tab >> 32));
                if (g_unique) break;
            }
            if (++h == buckets) h = 0;
            tab = table[h];
        }
    }
    if (o_inout) *o_inout = o;
}

typedef struct { size_t beg, end; } hole_t;
static int hole_cmp(const void *a, const void *b)
{
    size_t x = ((const hole_t *)a)->beg, y = ((const hole_t *)b)->beg;
    return x < y ? -1 : x > y;
}

/* npj.cpp:475-514, single worker: every worker's last block is filled up to
 * offsets[t]; the tail [offsets[t], block end) is a hole.  Tuples are moved
 * from the highest filled positions into the lowest holes until the filled
 * region is the dense prefix [0, J).  Returns J. */
size_t hjo_close_gaps(uint32_t *keys, uint32_t *vals, uint32_t *tabs,
                      const size_t *offsets, size_t count, size_t block_size)
{
    hole_t *holes = (hole_t *)malloc(count * sizeof(hole_t));
    for (size_t i = 0; i != count; ++i) {
        holes[i].beg = offsets[i];
        holes[i].end = (offsets[i] & ~(block_size - 1)) + block_size;
    }
    qsort(holes, count, sizeof(hole_t), hole_cmp);
    size_t l = 0, h = count - 1;
    size_t src = holes[h].end;
    while (l <= h) {
        size_t fill = src - holes[h].end;     /* filled tuples above hole h */
        if (fill == 0) {
            src = holes[h].beg;
            if (!h--) break;
            continue;
        }
        size_t hole = holes[l].end - holes[l].beg;
        if (hole == 0) { l++; continue; }
        size_t cnt = fill < hole ? fill : hole;
        size_t dst = holes[l].beg;
        holes[l].beg += cnt;
        src -= cnt;
        memmove(&keys[dst], &keys[src], cnt * sizeof(uint32_t));
        memmove(&vals[dst], &vals[src], cnt * sizeof(uint32_t));
        memmove(&tabs[dst], &tabs[src], cnt * sizeof(uint32_t));
    }
    free(holes);
    return src;
}

/* ------------------------------------------------------------------------ */
/* PHJ / CPRA per-partition operators                                        */
/* ------------------------------------------------------------------------ */

/* phj.cpp:577-603 (scalar form of 307-397): double hashing,
 * h1 = H(k*f0, B), step h2 = H(k*f1, B-1) + 1, insert at first empty. */
void hjo_phj_build(const uint32_t *keys, const uint32_t *vals, size_t size,
                   uint64_t *table, size_t buckets, const uint32_t factor[2],
                   uint32_t empty)
{
    for (size_t i = 0; i != buckets; ++i) table[i] = empty;
    for (size_t i = 0; i != size; ++i) {
        uint32_t k = keys[i];
        uint64_t pair = ((uint64_t)vals[i] << 32) | k;
        size_t h1 = hash_wide(k * factor[0], buckets);
        if ((uint32_t)table[h1] != empty) {
            size_t h2 = hash_wide(k * factor[1], buckets - 1) + 1;
            do {
                h1 += h2;
                if (h1 >= buckets) h1 -= buckets;
            } while ((uint32_t)table[h1] != empty);
        }
        table[h1] = pair;
    }
}

/* phj.cpp:605-647 (scalar form of 399-571); hjo_set_unique(1): first match only */
void hjo_phj_probe(const uint32_t *keys, const uint32_t *vals, size_t size,
                   const uint64_t *table, size_t buckets, const uint32_t factor[2],
                   uint32_t empty, hjo_result *agg, const hjo_output *out,
                   size_t *o_inout)
{
#if defined(__x86_64__)
    if (g_simd && !g_unique && !out && !o_inout && agg && buckets >= 3 && buckets < (1u << 31)) {
        hjo_phj_probe_avx512(keys, vals, size, table, buckets, factor, empty, agg);
        return;
    }
#endif
    size_t o = o_inout ? *o_inout : 0;
    for (size_t i = 0; i != size; ++i) {
        uint32_t k = keys[i];
        uint32_t v = vals[i];
        size_t h1 = hash_wide(k * factor[0], buckets);
        uint64_t t = table[h1];
        if ((uint32_t)t == empty) continue;
        size_t h2 = hash_wide(k * factor[1], buckets - 1) + 1;
        do {
            if ((uint32_t)t == k) {
                emit(agg, out, &o, k, v, (uint32_t)(t >> 32));
                if (g_unique) break;                   /* phj.cpp:635-637 */
            }
            h1 += h2;
            if (h1 >= buckets) h1 -= buckets;
            t = table[h1];
        } while ((uint32_t)t != empty);
    }
    if (o_inout) *o_inout = o;
}

/* ------------------------------------------------------------------------ */
/* thread harness                                                           */
/* ------------------------------------------------------------------------ */

typedef struct worker_ctx worker_ctx;
typedef void (*worker_fn)(worker_ctx *, int thread);
struct worker_ctx {
    int threads;
    pthread_barrier_t barrier;
    void *shared;
    worker_fn fn;
};
typedef struct { worker_ctx *ctx; int thread; } worker_arg;

static void *worker_main(void *p)
{
    worker_arg *a = (worker_arg *)p;
    /* thread i -> cpu i (SCATTER numbering, makefile:1 / npj.cpp:107-116);
     * best effort: ignored when the cpu is not in the allowed set. */
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(a->thread, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    a->ctx->fn(a->ctx, a->thread);
    return NULL;
}

static int run_workers(int threads, worker_fn fn, void *shared)
{
    worker_ctx ctx;
    ctx.threads = threads;
    ctx.shared = shared;
    ctx.fn = fn;
    pthread_barrier_init(&ctx.barrier, NULL, (unsigned)threads);
    pthread_t *ids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    worker_arg *args = (worker_arg *)malloc(sizeof(worker_arg) * (size_t)threads);
    for (int t = 0; t < threads; ++t) {
        args[t].ctx = &ctx;
        args[t].thread = t;
        if (pthread_create(&ids[t], NULL, worker_main, &args[t]) != 0) return -1;
    }
    for (int t = 0; t < threads; ++t) pthread_join(ids[t], NULL);
    pthread_barrier_destroy(&ctx.barrier);
    free(ids);
    free(args);
    return 0;
}

static void result_add(hjo_result *dst, const hjo_result *src)
{
    dst->count += src->count;
    dst->sum_keys += src->sum_keys;
    dst->sum_outer += src->sum_outer;
    dst->sum_inner += src->sum_inner;
}

/* ------------------------------------------------------------------------ */
/* NPJ: npj.cpp:769-927                                                      */
/* ------------------------------------------------------------------------ */

typedef struct {
    const uint32_t *ik, *iv, *ok, *ov;
    size_t inner, outer, buckets;
    uint32_t factor;
    uint64_t *table;
    const hjo_output *out;
    size_t *final_offsets;
    hjo_result *partial;
    double t[4];
} npj_shared;

static void npj_worker(worker_ctx *c, int thread)
{
    npj_shared *s = (npj_shared *)c->shared;
    size_t T = (size_t)c->threads, t = (size_t)thread;
    /* npj.cpp:783-788 */
    size_t ib = hjo_thread_beg(s->inner, 16, t, T), ie = hjo_thread_end(s->inner, 16, t, T);
    size_t ob = hjo_thread_beg(s->outer, 16, t, T), oe = hjo_thread_end(s->outer, 16, t, T);
    size_t tb = hjo_thread_beg(s->buckets, 16, t, T), te = hjo_thread_end(s->buckets, 16, t, T);
    pthread_barrier_wait(&c->barrier);
    if (thread == 0) s->t[0] = now_seconds();
    /* npj.cpp:865-868: set(table[range], empty = 0) */
    for (size_t i = tb; i != te; ++i) s->table[i] = 0;
    pthread_barrier_wait(&c->barrier);
    /* npj.cpp:871-877 */
    hjo_npj_build(&s->ik[ib], &s->iv[ib], ie - ib, s->table, s->buckets, s->factor, 0);
    pthread_barrier_wait(&c->barrier);
    if (thread == 0) s->t[1] = now_seconds();
    /* npj.cpp:882-901 */
    size_t o = 0;
    claim_first_block(s->out, &o);
    hjo_result agg = {0, 0, 0, 0};
    hjo_npj_probe(&s->ok[ob], &s->ov[ob], oe - ob, s->table, s->buckets, s->factor, 0,
                  &agg, s->out, &o);
    s->partial[thread] = agg;
    if (s->final_offsets) s->final_offsets[thread] = o;
    pthread_barrier_wait(&c->barrier);
    if (thread == 0) s->t[2] = now_seconds();
}

int hjo_npj(int threads,
            const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
            const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
            double load, uint32_t factor,
            hjo_result *res, const hjo_output *out, size_t *join_tuples_dense,
            hjo_timing *timing)
{
    if (threads < 1 || load <= 0.0 || load > 1.0) return -1;
    npj_shared s;
    memset(&s, 0, sizeof(s));
    s.ik = inner_keys; s.iv = inner_vals; s.ok = outer_keys; s.ov = outer_vals;
    s.inner = inner; s.outer = outer;
    /* npj.cpp:947: hash_buckets = inner / hash_table_load */
    s.buckets = (size_t)((double)inner / load);
    if (s.buckets <= inner) s.buckets = inner + 1;   /* a walk must find an empty bucket */
    s.factor = factor | 1u;
    s.table = (uint64_t *)malloc(s.buckets * sizeof(uint64_t));
    s.partial = (hjo_result *)calloc((size_t)threads, sizeof(hjo_result));
    s.final_offsets = out ? (size_t *)calloc((size_t)threads, sizeof(size_t)) : NULL;
    s.out = out;
    if (!s.table || !s.partial) return -1;
    if (run_workers(threads, npj_worker, &s) != 0) return -1;
    hjo_result total = {0, 0, 0, 0};
    for (int t = 0; t < threads; ++t) result_add(&total, &s.partial[t]);
    double t3 = s.t[2];
    if (out) {
        /* npj.cpp:903-915 */
        size_t dense = hjo_close_gaps(out->keys, out->outer_vals, out->inner_vals,
                                      s.final_offsets, (size_t)threads, out->block_size);
        t3 = now_seconds();
        if (join_tuples_dense) *join_tuples_dense = dense;
    }
    if (res) *res = total;
    if (timing) {
        timing->seconds = t3 - s.t[0];
        timing->seconds_phase[0] = s.t[1] - s.t[0];
        timing->seconds_phase[1] = s.t[2] - s.t[1];
        timing->seconds_phase[2] = t3 - s.t[2];
        timing->seconds_phase[3] = 0;
    }
    free(s.table); free(s.partial); free(s.final_offsets);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* pass planning shared by PHJ and CPRA: phj.cpp:1791-1808                   */
/* ------------------------------------------------------------------------ */

static size_t plan_passes(size_t partitions, size_t fanout[8])
{
    size_t passes = 0;
    if (partitions > 1000000) passes = 4;
    else if (partitions > 20000) passes = 3;
    else if (partitions > 400) passes = 2;
    else if (partitions > 10) passes = 1;
    size_t p;
    for (p = 0; p != passes; ++p)
        fanout[p] = (size_t)pow((double)partitions, 1.0 / (double)passes);
    fanout[p] = 1;
    if (passes) {
        size_t product = 1;
        for (p = 0; p != passes - 1; ++p) product *= fanout[p];
        fanout[p] = partitions / product;
    }
    return passes;
}

/* Local multi-pass partitioning of the ranges [ib,ie) / [ob,oe) held in
 * (*ik,*iv)/(*ok,*ov) with scratch twins: phj.cpp:1809-1863.
 * On return icounts / ocounts hold `return value` per-partition counts and the
 * in/out pointers have been swapped once per pass. */
static size_t local_passes(const size_t *fanout, hjo_rand32 *gen, uint32_t *factor_1st,
                           uint32_t **ik, uint32_t **iv, uint32_t **ik2, uint32_t **iv2,
                           size_t ib, size_t ie,
                           uint32_t **ok, uint32_t **ov, uint32_t **ok2, uint32_t **ov2,
                           size_t ob, size_t oe,
                           uint32_t **icounts, uint32_t **ocounts)
{
    size_t partitions = 1;
    uint32_t *ic = (uint32_t *)malloc(sizeof(uint32_t));
    uint32_t *oc = (uint32_t *)malloc(sizeof(uint32_t));
    ic[0] = (uint32_t)(ie - ib);
    oc[0] = (uint32_t)(oe - ob);
    for (size_t f = 0; fanout[f] != 1; ++f) {
        size_t fan = fanout[f];
        uint32_t *icn = (uint32_t *)malloc(partitions * fan * sizeof(uint32_t));
        uint32_t *ocn = (uint32_t *)malloc(partitions * fan * sizeof(uint32_t));
        uint32_t factor = hjo_rand32_next(gen) | 1u;       /* phj.cpp:1823 */
        if (f == 0 && factor_1st) *factor_1st = factor;
        size_t i = ib, o = ob;
        for (size_t p = 0; p != partitions; ++p) {
            size_t size = ic[p];
            hjo_histogram(&(*ik)[i], size, &icn[p * fan], factor, fan);
            hjo_partition(&(*ik)[i], &(*iv)[i], size, &icn[p * fan],
                          &(*ik2)[i], &(*iv2)[i], factor, fan);
            i += size;
            size = oc[p];
            hjo_histogram(&(*ok)[o], size, &ocn[p * fan], factor, fan);
            hjo_partition(&(*ok)[o], &(*ov)[o], size, &ocn[p * fan],
                          &(*ok2)[o], &(*ov2)[o], factor, fan);
            o += size;
        }
        free(ic); free(oc);
        ic = icn; oc = ocn;
        partitions *= fan;
        uint32_t *tmp;
        tmp = *ik; *ik = *ik2; *ik2 = tmp;
        tmp = *iv; *iv = *iv2; *iv2 = tmp;
        tmp = *ok; *ok = *ok2; *ok2 = tmp;
        tmp = *ov; *ov = *ov2; *ov2 = tmp;
    }
    *icounts = ic;
    *ocounts = oc;
    return partitions;
}

/* phj.cpp:1873-1876 */
static void draw_table_factors(hjo_rand32 *gen, uint32_t factors[2])
{
    do {
        factors[0] = hjo_rand32_next(gen) | 1u;
        factors[1] = hjo_rand32_next(gen) | 1u;
    } while (((factors[0] - factors[1]) & 3u) == 0);
}

/* phj.cpp:1886-1897: smallest e >= 1 that does not hash to partition 0 of the
 * outermost pass, so it cannot occur in the globally first partition (the only
 * one that can hold key 0). Falls back to 0 when there is no outer pass. */
static uint32_t first_partition_sentinel(uint32_t factor, size_t fanout)
{
    if (fanout < 2) return 0;
    uint32_t e = 0;
    size_t h;
    do {
        ++e;
        h = hash_wide(e * factor, fanout);
    } while (h == 0);
    return e;
}

/* ------------------------------------------------------------------------ */
/* PHJ: phj.cpp:1646-1949 (with the join loop 1869-1924 enabled)             */
/* ------------------------------------------------------------------------ */

typedef struct {
    uint32_t *ik[2], *iv[2], *ok[2], *ov[2];
    size_t inner, outer;
    double load;
    size_t hash_table_limit;
    uint32_t thread_factor, seed;
    uint32_t **icounts, **ocounts;      /* per-thread pass-1 histograms */
    hjo_result *partial;
    double t[4];
} phj_shared;

static void phj_worker(worker_ctx *c, int thread)
{
    phj_shared *s = (phj_shared *)c->shared;
    size_t T = (size_t)c->threads, t = (size_t)thread;
    uint32_t *ik = s->ik[0], *ik2 = s->ik[1], *iv = s->iv[0], *iv2 = s->iv[1];
    uint32_t *ok = s->ok[0], *ok2 = s->ok[1], *ov = s->ov[0], *ov2 = s->ov[1];
    hjo_rand32 gen;
    hjo_rand32_init(&gen, s->seed);                     /* phj.cpp:1672, 2127 */
    size_t ib = hjo_thread_beg(s->inner, 16, t, T), ie = hjo_thread_end(s->inner, 16, t, T);
    size_t ob = hjo_thread_beg(s->outer, 16, t, T), oe = hjo_thread_end(s->outer, 16, t, T);
    pthread_barrier_wait(&c->barrier);
    if (thread == 0) s->t[0] = now_seconds();
    if (T > 1) {
        /* phj.cpp:1715-1770: cross-thread pass, fan-out = #threads */
        uint32_t *ioff = (uint32_t *)malloc(T * 4 * sizeof(uint32_t));
        uint32_t *ooff = ioff + T, *iagg = ioff + 2 * T, *oagg = ioff + 3 * T;
        hjo_histogram(&ik[ib], ie - ib, s->icounts[thread], s->thread_factor, T);
        hjo_histogram(&ok[ob], oe - ob, s->ocounts[thread], s->thread_factor, T);
        pthread_barrier_wait(&c->barrier);
        size_t ni = hjo_interleave(s->icounts, ioff, iagg, T, t, T);
        size_t no = hjo_interleave(s->ocounts, ooff, oagg, T, t, T);
        assert(ni == s->inner && no == s->outer);
        (void)ni; (void)no;
        hjo_partition_shared(&ik[ib], &iv[ib], ie - ib, ioff, ik2, iv2, s->thread_factor, T);
        hjo_partition_shared(&ok[ob], &ov[ob], oe - ob, ooff, ok2, ov2, s->thread_factor, T);
        pthread_barrier_wait(&c->barrier);
        uint32_t *tmp;
        tmp = ik; ik = ik2; ik2 = tmp;  tmp = iv; iv = iv2; iv2 = tmp;
        tmp = ok; ok = ok2; ok2 = tmp;  tmp = ov; ov = ov2; ov2 = tmp;
        /* phj.cpp:1760-1766: my range = partition[thread] */
        ib = ob = 0;
        for (size_t u = 0; u != t; ++u) { ib += iagg[u]; ob += oagg[u]; }
        ie = ib + iagg[t];
        oe = ob + oagg[t];
        free(ioff);
        pthread_barrier_wait(&c->barrier);
    }
    if (thread == 0) s->t[1] = now_seconds();
    /* phj.cpp:1791-1808 */
    size_t fanout[8];
    size_t partitions = (ie - ib) / s->hash_table_limit;
    plan_passes(partitions, fanout);
    uint32_t factor_1st = 0;
    uint32_t *ic, *oc;
    partitions = local_passes(fanout, &gen, &factor_1st, &ik, &iv, &ik2, &iv2, ib, ie,
                              &ok, &ov, &ok2, &ov2, ob, oe, &ic, &oc);
    pthread_barrier_wait(&c->barrier);
    if (thread == 0) s->t[2] = now_seconds();
    /* phj.cpp:1869-1924 */
    double inverse_load = 1.0 / s->load;
    size_t max_buckets = 0;
    uint64_t *table = NULL;
    uint32_t factors[2];
    draw_table_factors(&gen, factors);
    hjo_result agg = {0, 0, 0, 0};
    size_t i = ib, o = ob;
    for (size_t p = 0; p != partitions; ++p) {
        uint32_t empty = 0;
        if (p == 0) {
            if (T == 1) empty = first_partition_sentinel(factor_1st, fanout[0]);
            else if (thread == 0) empty = first_partition_sentinel(s->thread_factor, T);
        }
        size_t size = ic[p];
        size_t buckets = (size_t)((double)size * inverse_load);
        if (buckets > max_buckets) {
            buckets = hjo_next_odd_prime(buckets);
            max_buckets = buckets;
            table = (uint64_t *)realloc(table, buckets * sizeof(uint64_t));
        } else if ((double)buckets * 1.2 > (double)max_buckets) {
            buckets = hjo_next_odd_prime(buckets);
        } else {
            buckets = max_buckets;
        }
        if (buckets < 3) buckets = 3;
        if (buckets > max_buckets) {
            max_buckets = buckets;
            table = (uint64_t *)realloc(table, buckets * sizeof(uint64_t));
        }
        hjo_phj_build(&ik[i], &iv[i], size, table, buckets, factors, empty);
        i += size;
        size = oc[p];
        hjo_phj_probe(&ok[o], &ov[o], size, table, buckets, factors, empty, &agg, NULL, NULL);
        o += size;
    }
    free(table); free(ic); free(oc);
    s->partial[thread] = agg;
    pthread_barrier_wait(&c->barrier);
    if (thread == 0) s->t[3] = now_seconds();
}

static uint32_t *dup_column(const uint32_t *src, size_t n)
{
    uint32_t *p = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    if (p && n) memcpy(p, src, n * sizeof(uint32_t));
    return p;
}

int hjo_phj(int threads,
            const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
            const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
            double load, size_t hash_table_limit, uint32_t thread_factor, uint32_t seed,
            hjo_result *res, hjo_timing *timing)
{
    if (threads < 1 || load <= 0.0 || load >= 1.0 || hash_table_limit == 0) return -1;
    if (inner >= (1ull << 32) || outer >= (1ull << 32)) return -1;  /* phj.cpp:1722-1727 */
    phj_shared s;
    memset(&s, 0, sizeof(s));
    /* [0] = working copy of the input, [1] = scratch twin (hj.h:1-72) */
    s.ik[0] = dup_column(inner_keys, inner); s.ik[1] = dup_column(inner_keys, inner);
    s.iv[0] = dup_column(inner_vals, inner); s.iv[1] = dup_column(inner_vals, inner);
    s.ok[0] = dup_column(outer_keys, outer); s.ok[1] = dup_column(outer_keys, outer);
    s.ov[0] = dup_column(outer_vals, outer); s.ov[1] = dup_column(outer_vals, outer);
    s.inner = inner; s.outer = outer; s.load = load;
    s.hash_table_limit = hash_table_limit;
    s.thread_factor = thread_factor | 1u;
    s.seed = seed;
    s.icounts = (uint32_t **)malloc(sizeof(uint32_t *) * (size_t)threads);
    s.ocounts = (uint32_t **)malloc(sizeof(uint32_t *) * (size_t)threads);
    for (int t = 0; t < threads; ++t) {
        s.icounts[t] = (uint32_t *)calloc((size_t)threads, sizeof(uint32_t));
        s.ocounts[t] = (uint32_t *)calloc((size_t)threads, sizeof(uint32_t));
    }
    s.partial = (hjo_result *)calloc((size_t)threads, sizeof(hjo_result));
    int rc = run_workers(threads, phj_worker, &s);
    hjo_result total = {0, 0, 0, 0};
    for (int t = 0; t < threads; ++t) result_add(&total, &s.partial[t]);
    if (res) *res = total;
    if (timing) {
        timing->seconds = s.t[3] - s.t[0];
        timing->seconds_phase[0] = s.t[1] - s.t[0];
        timing->seconds_phase[1] = s.t[2] - s.t[1];
        timing->seconds_phase[2] = s.t[3] - s.t[2];
        timing->seconds_phase[3] = 0;
    }
    for (int t = 0; t < threads; ++t) { free(s.icounts[t]); free(s.ocounts[t]); }
    free(s.icounts); free(s.ocounts); free(s.partial);
    for (int k = 0; k < 2; ++k) { free(s.ik[k]); free(s.iv[k]); free(s.ok[k]); free(s.ov[k]); }
    return rc;
}

/* ------------------------------------------------------------------------ */
/* CPRA: cpra2.cpp:1697-1986                                                 */
/* ------------------------------------------------------------------------ */

typedef struct {
    uint32_t *ik[2], *iv[2], *ok[2], *ov[2];
    size_t inner, outer;
    double load;
    size_t num_partitions;
    uint32_t seed;
    /* published after local partitioning (cpra2.cpp:1834-1839) */
    uint32_t **chunk_ik, **chunk_iv, **chunk_ok, **chunk_ov;
    uint32_t **build_icounts, **build_ocounts;
    size_t partitions;
    hjo_result *partial;
    double t[4];
    double copy_seconds;
} cpra_shared;

static void cpra_worker(worker_ctx *c, int thread)
{
    cpra_shared *s = (cpra_shared *)c->shared;
    size_t T = (size_t)c->threads, t = (size_t)thread;
    uint32_t *ik = s->ik[0], *ik2 = s->ik[1], *iv = s->iv[0], *iv2 = s->iv[1];
    uint32_t *ok = s->ok[0], *ok2 = s->ok[1], *ov = s->ov[0], *ov2 = s->ov[1];
    hjo_rand32 gen;
    hjo_rand32_init(&gen, s->seed);                      /* same seed in all threads, cpra2.cpp:2153 */
    size_t ib = hjo_thread_beg(s->inner, 16, t, T), ie = hjo_thread_end(s->inner, 16, t, T);
    size_t ob = hjo_thread_beg(s->outer, 16, t, T), oe = hjo_thread_end(s->outer, 16, t, T);
    pthread_barrier_wait(&c->barrier);
    if (thread == 0) s->t[0] = now_seconds();
    /* cpra2.cpp:1757-1827: partition OWN chunk only */
    size_t fanout[8];
    plan_passes(s->num_partitions, fanout);
    uint32_t factor_1st = 0;
    uint32_t *ic, *oc;
    size_t partitions = local_passes(fanout, &gen, &factor_1st, &ik, &iv, &ik2, &iv2, ib, ie,
                                     &ok, &ov, &ok2, &ov2, ob, oe, &ic, &oc);
    /* cpra2.cpp:1834-1840 */
    s->chunk_ik[thread] = &ik[ib]; s->chunk_iv[thread] = &iv[ib];
    s->chunk_ok[thread] = &ok[ob]; s->chunk_ov[thread] = &ov[ob];
    s->build_icounts[thread] = ic; s->build_ocounts[thread] = oc;
    if (thread == 0) s->partitions = partitions;
    pthread_barrier_wait(&c->barrier);
    if (thread == 0) s->t[1] = now_seconds();
    double inverse_load = 1.0 / s->load;
    uint32_t factors[2];
    draw_table_factors(&gen, factors);                   /* cpra2.cpp:1845-1849 */
    /* cpra2.cpp:1868-1872 */
    size_t par_start = (partitions / T) * t;
    size_t par_end = (partitions / T) * (t + 1);
    if (par_end > partitions || t == T - 1) par_end = partitions;
    /* cpra2.cpp:1875-1882: per-chunk running offsets of my first partition */
    size_t *ioff = (size_t *)calloc(T, sizeof(size_t));
    size_t *ooff = (size_t *)calloc(T, sizeof(size_t));
    for (size_t u = 0; u != T; ++u)
        for (size_t p = 0; p < par_start; ++p) {
            ioff[u] += s->build_icounts[u][p];
            ooff[u] += s->build_ocounts[u][p];
        }
    hjo_result agg = {0, 0, 0, 0};
    double copy = 0;
    for (size_t p = par_start; p != par_end; ++p) {
        size_t psize = 0;
        for (size_t u = 0; u != T; ++u) psize += s->build_icounts[u][p];
        uint32_t *rk = (uint32_t *)malloc((psize ? psize : 1) * sizeof(uint32_t));
        uint32_t *rv = (uint32_t *)malloc((psize ? psize : 1) * sizeof(uint32_t));
        /* cpra2.cpp:1891-1904: gather slice p of every chunk, in thread order */
        double c0 = now_seconds();
        size_t at = 0;
        for (size_t u = 0; u != T; ++u) {
            size_t n = s->build_icounts[u][p];
            memcpy(&rk[at], &s->chunk_ik[u][ioff[u]], n * sizeof(uint32_t));
            memcpy(&rv[at], &s->chunk_iv[u][ioff[u]], n * sizeof(uint32_t));
            ioff[u] += n; at += n;
        }
        copy += now_seconds() - c0;
        /* Sentinel: the reference tests thread_factor/#threads here although
         * CPRA has no thread-level pass (cpra2.cpp:1915-1919);
         * the restatement uses the outermost pass that really produced
         * partition 0. */
        uint32_t empty = (p == 0) ? first_partition_sentinel(factor_1st, fanout[0]) : 0;
        /* cpra2.cpp:1921-1923 */
        size_t buckets = hjo_next_odd_prime((size_t)((double)psize * inverse_load));
        if (buckets < 3) buckets = 3;
        uint64_t *table = (uint64_t *)malloc(buckets * sizeof(uint64_t));
        hjo_phj_build(rk, rv, psize, table, buckets, factors, empty);   /* cpra2.cpp:1938 */
        free(rk); free(rv);
        psize = 0;
        for (size_t u = 0; u != T; ++u) psize += s->build_ocounts[u][p];
        uint32_t *sk = (uint32_t *)malloc((psize ? psize : 1) * sizeof(uint32_t));
        uint32_t *sv = (uint32_t *)malloc((psize ? psize : 1) * sizeof(uint32_t));
        /* cpra2.cpp:1946-1959 */
        at = 0;
        for (size_t u = 0; u != T; ++u) {
            size_t n = s->build_ocounts[u][p];
            memcpy(&sk[at], &s->chunk_ok[u][ooff[u]], n * sizeof(uint32_t));
            memcpy(&sv[at], &s->chunk_ov[u][ooff[u]], n * sizeof(uint32_t));
            ooff[u] += n; at += n;
        }
        hjo_phj_probe(sk, sv, psize, table, buckets, factors, empty, &agg, NULL, NULL);
        free(sk); free(sv); free(table);
    }
    free(ioff); free(ooff);
    s->partial[thread] = agg;
    if (thread == 0) s->copy_seconds = copy;
    pthread_barrier_wait(&c->barrier);
    if (thread == 0) s->t[2] = now_seconds();
}

int hjo_cpra(int threads,
             const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
             const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
             double load, size_t num_partitions, uint32_t seed,
             hjo_result *res, hjo_timing *timing)
{
    if (threads < 1 || load <= 0.0 || load >= 1.0 || num_partitions == 0) return -1;
    if (inner >= (1ull << 32) || outer >= (1ull << 32)) return -1;
    cpra_shared s;
    memset(&s, 0, sizeof(s));
    s.ik[0] = dup_column(inner_keys, inner); s.ik[1] = dup_column(inner_keys, inner);
    s.iv[0] = dup_column(inner_vals, inner); s.iv[1] = dup_column(inner_vals, inner);
    s.ok[0] = dup_column(outer_keys, outer); s.ok[1] = dup_column(outer_keys, outer);
    s.ov[0] = dup_column(outer_vals, outer); s.ov[1] = dup_column(outer_vals, outer);
    s.inner = inner; s.outer = outer; s.load = load;
    s.num_partitions = num_partitions; s.seed = seed;
    size_t T = (size_t)threads;
    s.chunk_ik = (uint32_t **)calloc(T, sizeof(uint32_t *));
    s.chunk_iv = (uint32_t **)calloc(T, sizeof(uint32_t *));
    s.chunk_ok = (uint32_t **)calloc(T, sizeof(uint32_t *));
    s.chunk_ov = (uint32_t **)calloc(T, sizeof(uint32_t *));
    s.build_icounts = (uint32_t **)calloc(T, sizeof(uint32_t *));
    s.build_ocounts = (uint32_t **)calloc(T, sizeof(uint32_t *));
    s.partial = (hjo_result *)calloc(T, sizeof(hjo_result));
    int rc = run_workers(threads, cpra_worker, &s);
    hjo_result total = {0, 0, 0, 0};
    for (size_t t = 0; t < T; ++t) result_add(&total, &s.partial[t]);
    if (res) *res = total;
    if (timing) {
        timing->seconds = s.t[2] - s.t[0];
        timing->seconds_phase[0] = s.t[1] - s.t[0];
        timing->seconds_phase[1] = s.t[2] - s.t[1];
        timing->seconds_phase[2] = s.copy_seconds;       /* "copy:" cpra2.cpp:1984 */
        timing->seconds_phase[3] = 0;
    }
    for (size_t t = 0; t < T; ++t) { free(s.build_icounts[t]); free(s.build_ocounts[t]); }
    free(s.chunk_ik); free(s.chunk_iv); free(s.chunk_ok); free(s.chunk_ov);
    free(s.build_icounts); free(s.build_ocounts); free(s.partial);
    for (int k = 0; k < 2; ++k) { free(s.ik[k]); free(s.iv[k]); free(s.ok[k]); free(s.ov[k]); }
    return rc;
}

/* ------------------------------------------------------------------------ */
/* definition of the join (independent of every hash table above)            */
/* ------------------------------------------------------------------------ */

static int pair_cmp(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

void hjo_join_definition(const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
                         const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
                         hjo_result *res)
{
    hjo_result r = {0, 0, 0, 0};
    /* sort inner by key (key in the high half), then for each outer tuple sum
     * over the run of equal keys */
    uint64_t *pairs = (uint64_t *)malloc((inner ? inner : 1) * sizeof(uint64_t));
    for (size_t i = 0; i != inner; ++i)
        pairs[i] = ((uint64_t)inner_keys[i] << 32) | inner_vals[i];
    qsort(pairs, inner, sizeof(uint64_t), pair_cmp);
    for (size_t o = 0; o != outer; ++o) {
        uint32_t k = outer_keys[o];
        size_t lo = 0, hi = inner;
        while (lo < hi) {                       /* first pair with key >= k */
            size_t mid = lo + (hi - lo) / 2;
            if ((uint32_t)(pairs[mid] >> 32) < k) lo = mid + 1; else hi = mid;
        }
        for (; lo < inner && (uint32_t)(pairs[lo] >> 32) == k; ++lo) {
            r.count++;
            r.sum_keys += k;
            r.sum_outer += outer_vals[o];
            r.sum_inner += (uint32_t)pairs[lo];
        }
    }
    free(pairs);
    *res = r;
}
