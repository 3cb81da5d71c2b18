/*
 * hj_oracle_avx512.c — AVX-512 forms of the oracle's three hot operators, for the CPU baseline.
 *
 * TEST INFRASTRUCTURE ONLY (see hj_oracle.h).  bench.py's cpu_baseline leg times the CPU
 * restatement of run_hj; the reference's own inner loops are AVX-512 (phj.cpp:693-772
 * histogram, 1029-1231 partition, 399-571 probe), written with ICC-only KNC-heritage
 * intrinsics that do not compile here.  These are NOT translations of those
 * loops: they are written from the operators' definitions with standard <immintrin.h>
 * intrinsics, and tests/test_oracle_golden.py checks them bit for bit against the scalar
 * restatement and against the reference's own scalar operators on the committed fixtures.
 *
 *   histogram  16 lane-private counter rows (index p*16 + lane never collides inside a vector;
 *              the reference gets the same effect from 16 replicated 8-bit counters, phj.cpp:735-752)
 *   partition  16 partition ids per vector, same-partition lanes serialised with vpconflictd, masked
 *              gather / scatter of the line slots and of key and payload into per-partition 16-tuple
 *              write-combining lines, flushed with non-temporal 64-byte stores (the reference: lane-id
 *              scatter / gather-back, BUFFER_SIZE-tuple interleaved buffers + _mm512_stream_ps,
 *              phj.cpp:1099-1160); stable, so the output equals the scalar counting sort
 *   probe      16 chains advanced in lock step under a mask, 64-bit gathers of the buckets
 *              (the reference refills finished lanes instead, phj.cpp:427-435); aggregates only
 */
#include "hj_oracle.h"

#if defined(__AVX512F__) && defined(__AVX512CD__) && defined(__AVX512DQ__) && defined(__AVX512VL__)
#include <immintrin.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

int hjo_avx512_compiled(void) { return 1; }

/* 16 x mulhi32(x, n) for n < 2^32 */
static inline __m512i mulhi32(__m512i x, __m512i n)
{
    const __m512i even = _mm512_srli_epi64(_mm512_mul_epu32(x, n), 32);
    const __m512i odd = _mm512_mul_epu32(_mm512_srli_epi64(x, 32), n);
    return _mm512_mask_blend_epi32(0xAAAA, even, odd);
}

void hjo_histogram_avx512(const uint32_t *keys, size_t size, uint32_t *counts,
                          uint32_t factor, size_t partitions)
{
    if (partitions == 0) return;
    uint32_t *rows = (uint32_t *)aligned_alloc(64, partitions * 16 * sizeof(uint32_t));
    memset(rows, 0, partitions * 16 * sizeof(uint32_t));
    const __m512i f = _mm512_set1_epi32((int)factor), n = _mm512_set1_epi32((int)(uint32_t)partitions);
    const __m512i lane = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m512i one = _mm512_set1_epi32(1);
    size_t i = 0;
    for (; i + 16 <= size; i += 16) {
        const __m512i k = _mm512_loadu_si512((const void *)(keys + i));
        const __m512i p = mulhi32(_mm512_mullo_epi32(k, f), n);
        const __m512i idx = _mm512_add_epi32(_mm512_slli_epi32(p, 4), lane);
        const __m512i c = _mm512_i32gather_epi32(idx, rows, 4);
        _mm512_i32scatter_epi32(rows, idx, _mm512_add_epi32(c, one), 4);
    }
    for (size_t p = 0; p != partitions; ++p)
        counts[p] = _mm512_reduce_add_epi32(_mm512_load_si512((const void *)(rows + p * 16)));
    for (; i != size; ++i)
        counts[(uint32_t)(((uint64_t)(uint32_t)(keys[i] * factor) * partitions) >> 32)]++;
    free(rows);
}

/* Appends the tuples to the partitions' cursors `offsets` (advanced), through one
 * write-combining line per partition and column.  Line w of the output covers positions
 * [16w, 16w + 16); a line leaves with a non-temporal store when its last slot is filled, or
 * with a masked store when it is only partly ours (first line of a range that starts in the
 * middle of a line, last lines at the end). */
static int g_vector_scatter = 0;
/* 0: 16 partition ids per vector, tuples moved with scalar stores into the write-combining lines (the faster form on
 * the CPUs measured: 0.46 against 0.30 Gtuples/s for the whole PHJ, 8 threads of a Xeon with slow scatters);
 * 1: the reference's shape, conflict-serialised masked gather / scatter (phj.cpp:1099-1160).  Same output. */
void hjo_avx512_vector_scatter(int on) { g_vector_scatter = on ? 1 : 0; }

static void partition_wc(const uint32_t *keys, const uint32_t *vals, size_t size, size_t *offsets,
                         uint32_t *keys_out, uint32_t *vals_out, uint32_t factor, size_t partitions)
{
    if (partitions == 0 || ((((uintptr_t)keys_out ^ (uintptr_t)vals_out) >> 2) & 15) != 0) {
        /* the two output columns do not share their position inside a line: plain definition */
        for (size_t i = 0; i != size; ++i) {
            const size_t o = offsets[(uint32_t)(((uint64_t)(uint32_t)(keys[i] * factor) * partitions) >> 32)]++;
            keys_out[o] = keys[i]; vals_out[o] = vals[i];
        }
        return;
    }
    uint32_t *bk = (uint32_t *)aligned_alloc(64, partitions * 16 * sizeof(uint32_t));
    uint32_t *bv = (uint32_t *)aligned_alloc(64, partitions * 16 * sizeof(uint32_t));
    uint16_t *first = (uint16_t *)malloc(partitions * sizeof(uint16_t));   /* first slot of the open line that is ours */
    /* lines are 64-byte lines of MEMORY: slot of output position o = (o + shift) & 15 */
    const size_t shift = ((uintptr_t)keys_out >> 2) & 15;
    for (size_t p = 0; p != partitions; ++p) first[p] = (uint16_t)((offsets[p] + shift) & 15);
    const __m512i f = _mm512_set1_epi32((int)factor), n = _mm512_set1_epi32((int)(uint32_t)partitions);
    uint32_t part[16] __attribute__((aligned(64)));
    size_t i = 0;
#define HJO_PUT(P, K, V)                                                                         \
    do {                                                                                         \
        const size_t p_ = (P);                                                                   \
        const size_t o_ = offsets[p_]++;                                                         \
        const size_t s_ = (o_ + shift) & 15;                                                     \
        bk[p_ * 16 + s_] = (K); bv[p_ * 16 + s_] = (V);                                          \
        if (s_ == 15) {                                                                          \
            const ptrdiff_t base_ = (ptrdiff_t)o_ - 15;      /* may lie before keys_out: masked */ \
            const __m512i lk_ = _mm512_load_si512((const void *)(bk + p_ * 16));                 \
            const __m512i lv_ = _mm512_load_si512((const void *)(bv + p_ * 16));                 \
            if (first[p_] == 0) {                                                                \
                _mm512_stream_si512((__m512i *)(keys_out + base_), lk_);                         \
                _mm512_stream_si512((__m512i *)(vals_out + base_), lv_);                         \
            } else {                                                                             \
                const __mmask16 m_ = (__mmask16)(0xFFFFu << first[p_]);                          \
                _mm512_mask_storeu_epi32(keys_out + base_, m_, lk_);                             \
                _mm512_mask_storeu_epi32(vals_out + base_, m_, lv_);                             \
                first[p_] = 0;                                                                   \
            }                                                                                    \
        }                                                                                        \
    } while (0)
    if (g_vector_scatter) {
        /* Vector main loop, the shape of the reference's (phj.cpp:1099-1160): 16 partition ids per vector; lanes that
         * hit the same partition are serialised (the reference writes lane ids and reads them back, phj.cpp:1099-1102;
         * AVX-512CD has vpconflictd: a lane is ready when no EARLIER pending lane shares its partition, which also keeps
         * the sort stable); the ready lanes gather their line slots, scatter key and payload into the write-combining
         * lines and bump the slots with one more scatter; a line whose last slot was just filled leaves with
         * non-temporal stores (phj.cpp:1124-1160).  slot16[p] = slot of the partition's next tuple inside its line. */
        uint32_t *slot16 = (uint32_t *)aligned_alloc(64, ((partitions + 15) & ~(size_t)15) * sizeof(uint32_t));
        for (size_t p = 0; p != partitions; ++p) slot16[p] = (uint32_t)((offsets[p] + shift) & 15);
        const __m512i fifteen = _mm512_set1_epi32(15), one = _mm512_set1_epi32(1);
        for (; i + 16 <= size; i += 16) {
            const __m512i k = _mm512_loadu_si512((const void *)(keys + i));
            const __m512i v = _mm512_loadu_si512((const void *)(vals + i));
            const __m512i p = mulhi32(_mm512_mullo_epi32(k, f), n);
            __mmask16 pending = 0xFFFF;
            do {
                /* conflict bits of lane l: earlier lanes with the same partition; only pending ones still block it */
                const __m512i conf = _mm512_and_epi32(_mm512_conflict_epi32(p), _mm512_set1_epi32((int)pending));
                const __mmask16 ready = _mm512_mask_cmpeq_epi32_mask(pending, conf, _mm512_setzero_si512());
                const __m512i slot = _mm512_mask_i32gather_epi32(_mm512_setzero_si512(), ready, p, slot16, 4);
                const __m512i at = _mm512_add_epi32(_mm512_slli_epi32(p, 4), slot);
                _mm512_mask_i32scatter_epi32(bk, ready, at, k, 4);
                _mm512_mask_i32scatter_epi32(bv, ready, at, v, 4);
                _mm512_mask_i32scatter_epi32(slot16, ready, p, _mm512_and_epi32(_mm512_add_epi32(slot, one), fifteen), 4);
                __mmask16 full = _mm512_mask_cmpeq_epi32_mask(ready, slot, fifteen);
                if (full) {
                    _mm512_store_si512((void *)part, p);
                    while (full) {
                        const int l = __builtin_ctz(full);
                        full &= (__mmask16)(full - 1);
                        const size_t p_ = part[l];
                        /* tuples of this partition so far: the line's last slot belongs to position offsets[p] + 15 - first... */
                        const size_t o_ = offsets[p_] + (size_t)(15 - ((offsets[p_] + shift) & 15));   /* position of slot 15 */
                        const ptrdiff_t base_ = (ptrdiff_t)o_ - 15;      /* may lie before keys_out: masked */
                        const __m512i lk_ = _mm512_load_si512((const void *)(bk + p_ * 16));
                        const __m512i lv_ = _mm512_load_si512((const void *)(bv + p_ * 16));
                        if (first[p_] == 0) {
                            _mm512_stream_si512((__m512i *)(keys_out + base_), lk_);
                            _mm512_stream_si512((__m512i *)(vals_out + base_), lv_);
                        } else {
                            const __mmask16 m_ = (__mmask16)(0xFFFFu << first[p_]);
                            _mm512_mask_storeu_epi32(keys_out + base_, m_, lk_);
                            _mm512_mask_storeu_epi32(vals_out + base_, m_, lv_);
                            first[p_] = 0;
                        }
                        offsets[p_] = o_ + 1;                            /* the next line starts here */
                    }
                }
                pending &= (__mmask16)~ready;
            } while (pending);
        }
        /* `offsets` was advanced line by line only: add what sits in the open lines */
        for (size_t p = 0; p != partitions; ++p) {
            const size_t s0 = (offsets[p] + shift) & 15;
            offsets[p] += (size_t)((slot16[p] - s0) & 15);
        }
        free(slot16);
    } else {
        for (; i + 16 <= size; i += 16) {
            const __m512i k = _mm512_loadu_si512((const void *)(keys + i));
            _mm512_store_si512((void *)part, mulhi32(_mm512_mullo_epi32(k, f), n));
            for (int l = 0; l != 16; ++l) HJO_PUT(part[l], keys[i + l], vals[i + l]);
        }
    }
    for (; i != size; ++i)
        HJO_PUT((uint32_t)(((uint64_t)(uint32_t)(keys[i] * factor) * partitions) >> 32), keys[i], vals[i]);
#undef HJO_PUT
    /* open lines */
    for (size_t p = 0; p != partitions; ++p) {
        const size_t end = offsets[p], s = (end + shift) & 15;
        if (s > first[p]) {
            const __mmask16 m = (__mmask16)((0xFFFFu << first[p]) & (0xFFFFu >> (16 - s)));
            const ptrdiff_t base = (ptrdiff_t)end - (ptrdiff_t)s;
            _mm512_mask_storeu_epi32(keys_out + base, m, _mm512_load_si512((const void *)(bk + p * 16)));
            _mm512_mask_storeu_epi32(vals_out + base, m, _mm512_load_si512((const void *)(bv + p * 16)));
        }
    }
    _mm_sfence();
    free(bk); free(bv); free(first);
}

void hjo_partition_avx512(const uint32_t *keys, const uint32_t *vals, size_t size,
                          const uint32_t *counts, uint32_t *keys_out, uint32_t *vals_out,
                          uint32_t factor, size_t partitions)
{
    size_t *offsets = (size_t *)malloc((partitions ? partitions : 1) * sizeof(size_t));
    size_t acc = 0;
    for (size_t p = 0; p != partitions; ++p) { offsets[p] = acc; acc += counts[p]; }
    partition_wc(keys, vals, size, offsets, keys_out, vals_out, factor, partitions);
    free(offsets);
}

void hjo_partition_shared_avx512(const uint32_t *keys, const uint32_t *vals, size_t size,
                                 uint32_t *offsets32, uint32_t *keys_out, uint32_t *vals_out,
                                 uint32_t factor, size_t partitions)
{
    size_t *offsets = (size_t *)malloc((partitions ? partitions : 1) * sizeof(size_t));
    for (size_t p = 0; p != partitions; ++p) offsets[p] = offsets32[p];
    partition_wc(keys, vals, size, offsets, keys_out, vals_out, factor, partitions);
    for (size_t p = 0; p != partitions; ++p) offsets32[p] = (uint32_t)offsets[p];
    free(offsets);
}

/* aggregates only (count + 3 sums); buckets < 2^31 */
void hjo_phj_probe_avx512(const uint32_t *keys, const uint32_t *vals, size_t size,
                          const uint64_t *table, size_t buckets, const uint32_t factor[2],
                          uint32_t empty, hjo_result *agg)
{
    const __m512i f0 = _mm512_set1_epi32((int)factor[0]), f1 = _mm512_set1_epi32((int)factor[1]);
    const __m512i nb = _mm512_set1_epi32((int)(uint32_t)buckets), nb1 = _mm512_set1_epi32((int)(uint32_t)(buckets - 1));
    const __m512i vempty = _mm512_set1_epi32((int)empty), one = _mm512_set1_epi32(1);
    const __m512i lo_idx = _mm512_setr_epi32(0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30);
    const __m512i hi_idx = _mm512_setr_epi32(1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31);
    __m512i sk = _mm512_setzero_si512(), so = sk, si = sk;      /* 8 x u64 each */
    uint64_t count = 0;
    size_t i = 0;
    for (; i + 16 <= size; i += 16) {
        const __m512i k = _mm512_loadu_si512((const void *)(keys + i));
        const __m512i v = _mm512_loadu_si512((const void *)(vals + i));
        __m512i h = mulhi32(_mm512_mullo_epi32(k, f0), nb);
        const __m512i step = _mm512_add_epi32(mulhi32(_mm512_mullo_epi32(k, f1), nb1), one);
        __mmask16 live = 0xFFFF;
        while (live) {
            /* 16 buckets of 8 bytes: two gathers of 8 */
            const __m512i b0 = _mm512_mask_i32gather_epi64(_mm512_setzero_si512(), (__mmask8)live,
                                                           _mm512_castsi512_si256(h), table, 8);
            const __m512i b1 = _mm512_mask_i32gather_epi64(_mm512_setzero_si512(), (__mmask8)(live >> 8),
                                                           _mm512_extracti64x4_epi64(h, 1), table, 8);
            const __m512i tk = _mm512_permutex2var_epi32(b0, lo_idx, b1);   /* low words  = keys     */
            const __m512i tv = _mm512_permutex2var_epi32(b0, hi_idx, b1);   /* high words = payloads */
            live = _mm512_mask_cmpneq_epi32_mask(live, tk, vempty);
            const __mmask16 hit = _mm512_mask_cmpeq_epi32_mask(live, tk, k);
            if (hit) {
                count += (uint64_t)__builtin_popcount(hit);
                const __m512i mk = _mm512_maskz_mov_epi32(hit, k), mo = _mm512_maskz_mov_epi32(hit, v),
                              mi = _mm512_maskz_mov_epi32(hit, tv);
                sk = _mm512_add_epi64(sk, _mm512_add_epi64(_mm512_cvtepu32_epi64(_mm512_castsi512_si256(mk)),
                                                           _mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(mk, 1))));
                so = _mm512_add_epi64(so, _mm512_add_epi64(_mm512_cvtepu32_epi64(_mm512_castsi512_si256(mo)),
                                                           _mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(mo, 1))));
                si = _mm512_add_epi64(si, _mm512_add_epi64(_mm512_cvtepu32_epi64(_mm512_castsi512_si256(mi)),
                                                           _mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(mi, 1))));
            }
            /* h = (h + step) mod buckets, both < buckets < 2^31 */
            h = _mm512_add_epi32(h, step);
            h = _mm512_mask_sub_epi32(h, _mm512_cmpge_epu32_mask(h, nb), h, nb);
        }
    }
    agg->count += count;
    agg->sum_keys += (uint64_t)_mm512_reduce_add_epi64(sk);
    agg->sum_outer += (uint64_t)_mm512_reduce_add_epi64(so);
    agg->sum_inner += (uint64_t)_mm512_reduce_add_epi64(si);
    /* tail: scalar definition */
    for (; i != size; ++i) {
        const uint32_t k = keys[i];
        size_t h1 = (size_t)(((uint64_t)(uint32_t)(k * factor[0]) * buckets) >> 32);
        uint64_t t = table[h1];
        if ((uint32_t)t == empty) continue;
        const size_t h2 = (size_t)(((uint64_t)(uint32_t)(k * factor[1]) * (buckets - 1)) >> 32) + 1;
        do {
            if ((uint32_t)t == k) {
                agg->count++; agg->sum_keys += k; agg->sum_outer += vals[i]; agg->sum_inner += (uint32_t)(t >> 32);
            }
            h1 += h2;
            if (h1 >= buckets) h1 -= buckets;
            t = table[h1];
        } while ((uint32_t)t != empty);
    }
}

#else  /* built without AVX-512: the scalar restatement is all there is */
int hjo_avx512_compiled(void) { return 0; }
void hjo_avx512_vector_scatter(int on) { (void)on; }
#endif
