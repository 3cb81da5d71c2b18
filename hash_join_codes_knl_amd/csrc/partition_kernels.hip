// partition_kernels.hip — K4 histogram, K5 prefix/plan, K6 scatter for gfx950.
//
// Replaces histogram()/interleave()/partition() (phj.cpp:693-772, 1263-1291,
// 1029-1231; cpra2.cpp:801-1075).  Design (not a translation):
//   * K4 reads the key column once with 16-byte loads and counts BOTH pass
//     levels at once (bin = p1*F2 + p2) in a per-workgroup LDS histogram
//     (ds_add_u32), flushed with one global atomic per non-empty bin.  The
//     reference re-reads the keys before every pass (phj.cpp:1819-1863).
//   * K5 turns the counts into absolute 64-bit offsets, atomic write cursors
//     and the tile/work-item prefix tables the later kernels walk, all on the
//     device (no host round trip between kernels).
//   * K6 counting-sorts one 8192-tuple tile by partition inside LDS (the
//     reference's per-partition write-combining buffers, phj.cpp:1115-1160,
//     become one LDS tile), claims one contiguous output run per partition
//     with a single returning atomic on the partition's cursor, and streams
//     the tile out so that consecutive lanes write consecutive addresses.
//
// Unaligned segment starts are handled by reading whole aligned 16-byte
// vectors and masking the lanes outside the segment: an aligned vector that
// overlaps a valid element never leaves that element's page.
#include "hj_device.hpp"
#include "hj_internal.hpp"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

// --------------------------------------------------------------------------
// How K6's stores leave the CU (HJ_K6_STORE, a build-time choice; tools/build_variant.py <name> -DHJ_K6_STORE=n):
//   0  plain stores (write-back in the XCD's L2)
//   1  non-temporal stores (global_store ... nt) - THE PRODUCT
//   2  system-scope stores (two relaxed 8-byte __hip_atomic_store per 16 bytes: global_store_dwordx2 ... sc0 sc1, written through)
//   3 / 4  timing experiments only: 16-byte stores non-temporal and 8-byte stores plain / the other way round
// Round 5 (DESIGN section 3 "Round 5", profiles/r05_*.txt): with plain stores a K6 launch LOSES STORES - output slots keep what
// the previous use of the buffer left there - in 1.5 of 10^4 steps of the multi-GPU slice pipeline (13 wrong steps in 84 000),
// i.e. when kernels and copies of ANOTHER stream start and end beside it; never on one stream (0 in 15 000), never with
// partitioning and joins serialised (0 in 15 000).  Option "audit" named the kernel: the sums of pass 2's output differ from its
// input's with nothing misplaced and the count intact.  The same machine code with the nt bit on its stores: 0 wrong steps in
// 73 000; written through (sc0 sc1): 0 in 15 000.  Rounds 3-4 saw the same signature at a far higher rate with a private segment in
// the kernel and studied it with -DHJ_SCRATCH_EXPERIMENT=n variants of this file (profiles/r04_scratch_repro.txt; the variants
// are in the history before round 5, not here).  What is lost is data that sits dirty in an XCD's L2 while another queue's
// kernel boundary writes back and invalidates that L2; non-temporal stores do not linger there.
// --------------------------------------------------------------------------
#ifndef HJ_K6_STORE
#define HJ_K6_STORE 1
#endif
__device__ __forceinline__ void k6_store16(u64 *p, uint4 v)
{
#if HJ_K6_STORE == 1 || HJ_K6_STORE == 3
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    v4u_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4u_t *>(p));
#elif HJ_K6_STORE == 2
    __hip_atomic_store(p, (u64)v.x | ((u64)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p + 1, (u64)v.z | ((u64)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    *reinterpret_cast<uint4 *>(p) = v;
#endif
}
// 8-byte stores write PARTIAL lines (the < 16 tail tuples of a pass-2 run, the edges of a pass-1 run).  Non-temporal they cost
// 0.15-0.4 ms per pass (a partial line that leaves the L2 at once is a read-modify-write at the memory side; plain, the L2
// waits for the rest of the line), while the 16-byte whole-line stores are FASTER non-temporal than plain
// (profiles/r05_ab_stores.txt, ms pass 1 / pass 2 at 64 M x 1 G in the same allocations: all plain 2.99 / 3.07, 16-byte nt
// 2.93 / 2.96, all nt 3.07 / 3.13, only 8-byte nt 3.37 / 3.20).  So: whole lines always non-temporal; partial lines non-temporal
// unless the launch is SOLO (ScatterArgs::nt_partial = 0 selects the NTP = false instance: a blocking join of a context whose option "solo" says that nothing else
// runs on the device beside it - the condition under which no store was ever lost: 0 wrong steps in 15 000 on one stream with
// every store plain).
template <bool NT>
__device__ __forceinline__ void k6_store8(u64 *p, u64 v)
{
#if HJ_K6_STORE == 1
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
#elif HJ_K6_STORE == 4
    __builtin_nontemporal_store(v, p);
#elif HJ_K6_STORE == 2
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    *p = v;
#endif
}

// --------------------------------------------------------------------------
// K4: fused two-level histogram + per-range pass-1 counts.
// A workgroup walks whole ranges (Pass1Geom); for every range it leaves the
// F1 pass-1 counts of that range in range_counts[range][F1] (non-temporal stores like every store of the library, hj_device.hpp), and
// it accumulates the fused (p1,p2) histogram of everything it saw in LDS, flushed
// once at the end with one global atomic per non-empty bin.
// --------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void hist2_kernel(
    const uint32_t *__restrict__ keys, Pass1Geom geom_arg,
    uint32_t f1, uint32_t F1, uint32_t f2, uint32_t F2,
    u64 *__restrict__ counts, uint32_t *__restrict__ range_counts, uint32_t *work_counter, const u64 *__restrict__ dyn)
{
    // a device-planned group (hjgpu_api.hip phj_grouped_device): where the relation starts inside `keys` and how many rows it has are
    // known on the device only; the ranges per chunk and the tile stay those of the capacity the workspace was planned for
    Pass1Geom geom = geom_arg;
    if (dyn) { keys += dyn[0]; geom.n = dyn[1]; geom.part = (geom.n / geom.chunks) & ~(u64)15; }
    extern __shared__ uint32_t lds_hist[];          // [P] fused, then [F1] per-range
    const uint32_t P = F1 * F2;
    uint32_t *range_hist = lds_hist + P;
    const uint32_t Rc = geom.ranges_per_chunk;
    const uint32_t chunk = blockIdx.y;                  // one grid row per chunk: a workgroup never mixes chunks
    counts += (u64)chunk * P;
    range_counts += (u64)chunk * Rc * F1;
    const u64 cb = geom.beg(chunk), ce = geom.beg(chunk + 1);
    const uint4 *__restrict__ k4 = reinterpret_cast<const uint4 *>(keys - geom.align);

    for (uint32_t i = threadIdx.x; i < P + F1; i += BLOCK) lds_hist[i] = 0;     // fused histogram + range_hist

    // ranges are claimed from a per-chunk ticket counter (see K6: whoever runs, works); the next
    // ticket is requested while the current range is counted
    __shared__ uint32_t next_range;
    // Skew: the fullest bin of the fused histogram so far (count << 32 | bin; counts only grow, so an atomic max
    // over the range-end sweeps needs no reset).  When it holds more than 1/8 of what this workgroup has counted,
    // the next range serves that bin's lanes with ONE LDS add per wave instruction (ballot + popcount) instead of
    // letting up to 64 lanes queue on one LDS word: Zipf(2.0) 1.78 -> 0.9x ms; uniform keys never take the path.
    __shared__ u64 fullest_bin;
    u64 counted = 0;                                    // tuples this workgroup has histogrammed (uniform)
    uint32_t *ticket = work_counter + chunk;
    if (threadIdx.x == 0) { next_range = atomicAdd(ticket, 1u); fullest_bin = 0; }
    __syncthreads();
    for (;;) {
        const uint32_t r = next_range;
        if (r >= Rc) break;
        const u64 fullest = fullest_bin;
        const uint32_t hot_bin = (uint32_t)fullest;
        const bool hot = F2 > 1 && counted > 0 && (fullest >> 32) * 8 > counted;
        uint32_t upcoming = 0;
        if (threadIdx.x == 0) upcoming = atomicAdd(ticket, 1u);
        const uint32_t j = r;
        const u64 gb = geom.align + cb, ge = geom.align + ce;
        const u64 tiles = hj_tiles_of(cb, ce, geom.align, geom.tile);
        const u64 t_beg = tiles * j / Rc, t_end = tiles * (j + 1) / Rc;
        const u64 g_lo = (gb & ~3ull) + t_beg * geom.tile;
        const u64 g_hi = min(ge, (gb & ~3ull) + t_end * geom.tile);

        // F2 > 1: ONE LDS add per key, into the fused histogram; the range's pass-1 counts are the growth of
        // the fused rows' sums over the range (range_hist keeps the sums seen so far).  The second add per
        // key (a private pass-1 histogram per range) made K4 LDS-atomic-bound: 0.80 ms for a sweep that the
        // memory system delivers in 0.64 ms.  F2 == 1: the pass-1 histogram is the only one.
        if (F2 == 1) {
            for (uint32_t i = threadIdx.x; i < F1; i += BLOCK) range_hist[i] = 0;
            __syncthreads();
        }
        auto count_range = [&](auto part1, auto hot_tag) {
            constexpr bool HOT = decltype(hot_tag)::value;
            constexpr int U = 4;                       // key vectors per lane and batch; two batches in flight
            const u64 step = (u64)BLOCK * 4 * U;
            // A batch is WHOLE when all its BLOCK * 4 * U keys lie inside the range and the chunk (uniform test): no
            // per-vector and per-key predicates then, i.e. no exec-mask bookkeeping around the loads and LDS adds
            // (all batches of a range but the chunk's first and the range's last).
            auto whole = [&](u64 base) { return base >= gb && base + step <= g_hi && base + step <= ge; };
            auto fetch = [&](u64 g0, uint4 (&kv)[U], bool all) {
                if (all) {
#pragma unroll
                    for (int u = 0; u < U; ++u) kv[u] = hj_load_nt(k4 + ((g0 + (u64)u * BLOCK * 4) >> 2));   // read once: 0.75 -> 0.70 ms
                    return;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const u64 g = g0 + (u64)u * BLOCK * 4;
                    kv[u] = make_uint4(0, 0, 0, 0);
                    if (g < g_hi) kv[u] = k4[g >> 2];
                }
            };
            auto add_key = [&](uint32_t key) {
                const uint32_t p1 = part1(key);
                if (HOT) {
                    const uint32_t bin = p1 * F2 + hj_hash(key, f2, F2);
                    const bool is_hot = bin == hot_bin;
                    const u64 m = __ballot(is_hot);             // over the lanes with a valid key here
                    if (!is_hot) atomicAdd(&lds_hist[bin], 1u);
                    else if (hj_lane() == (uint32_t)__builtin_ctzll(m)) atomicAdd(&lds_hist[hot_bin], (uint32_t)__popcll(m));
                } else if (F2 > 1) atomicAdd(&lds_hist[p1 * F2 + hj_hash(key, f2, F2)], 1u);
                else atomicAdd(&range_hist[p1], 1u);
            };
            auto count = [&](u64 g0, const uint4 (&kv)[U], bool all) {
                if (all) {
#pragma unroll
                    for (int u = 0; u < U; ++u) { add_key(kv[u].x); add_key(kv[u].y); add_key(kv[u].z); add_key(kv[u].w); }
                    return;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const u64 g = g0 + (u64)u * BLOCK * 4;
                    if (g >= g_hi) break;
                    const uint32_t kk[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w};
                    const bool full = (g >= gb) && (g + 4 <= ge);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (full || (g + e >= gb && g + e < ge)) add_key(kk[e]);
                }
            };
            // the next batch's loads are issued before the current batch is counted: the lane always has
            // 4-8 loads outstanding (one workgroup per CU at fan-outs whose histogram takes > 72 KiB of LDS)
            uint4 ka[U], kb[U];
            u64 base = g_lo;                                            // first key of the batch (uniform)
            const u64 mine = (u64)threadIdx.x * 4;
            bool wa = whole(base), wb;
            fetch(base + mine, ka, wa);
            while (base < g_hi) {
                wb = whole(base + step);
                fetch(base + step + mine, kb, wb);
                count(base + mine, ka, wa);
                base += step;
                if (base >= g_hi) break;
                wa = whole(base + step);
                fetch(base + step + mine, ka, wa);
                count(base + mine, kb, wb);
                base += step;
            }
        };
        if (t_end > t_beg) {
            // H(key, f, 2^k) = (key * f) >> (32 - k): same function, one multiply less
            if (F1 > 1 && (F1 & (F1 - 1)) == 0) {
                const uint32_t sh = 32 - (uint32_t)__builtin_ctz(F1);
                auto part = [&](uint32_t k) { return (k * f1) >> sh; };
                if (hot) count_range(part, std::true_type()); else count_range(part, std::false_type());
            } else {
                auto part = [&](uint32_t k) { return hj_hash(k, f1, F1); };
                if (hot) count_range(part, std::true_type()); else count_range(part, std::false_type());
            }
            counted += g_hi - g_lo;
        }
        __syncthreads();
        uint32_t *__restrict__ rc = range_counts + (u64)r * F1;
        if (F2 > 1) {
            // one wave per fused row: sum of the row now minus the sum after the previous range
            u64 best = 0;                                           // count << 32 | bin of the fullest bin this lane saw
            for (uint32_t p1 = threadIdx.x >> 6; p1 < F1; p1 += BLOCK / 64) {
                uint32_t sum = 0;
                for (uint32_t p2 = hj_lane(); p2 < F2; p2 += 64) {
                    const uint32_t v = lds_hist[p1 * F2 + p2];
                    sum += v;
                    best = max(best, ((u64)v << 32) | (p1 * F2 + p2));
                }
                sum = (uint32_t)wave_reduce_sum((u64)sum);
                if (hj_lane() == 0) { hj_store(&rc[p1], sum - range_hist[p1]); range_hist[p1] = sum; }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) best = max(best, (u64)__shfl_down((unsigned long long)best, d, 64));
            if (hj_lane() == 0 && (best >> 32) * 8 > counted) atomicMax(reinterpret_cast<unsigned long long *>(&fullest_bin), (unsigned long long)best);
        } else {
            for (uint32_t i = threadIdx.x; i < F1; i += BLOCK) {
                const uint32_t v = range_hist[i];
                hj_store(&rc[i], v);
                if (v) atomicAdd(&lds_hist[i], v);              // single pass: fused == pass-1 histogram
            }
        }
        if (threadIdx.x == 0) next_range = upcoming;            // everybody read the old value before the first barrier above
        __syncthreads();
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < P; i += BLOCK) {
        const uint32_t v = lds_hist[i];
        if (v) atomicAdd(&counts[i], (u64)v);
    }
}

int hj_launch_hist2(const uint32_t *keys, const Pass1Geom &geom,
                    uint32_t f1, uint32_t F1, uint32_t f2, uint32_t F2,
                    u64 *counts, uint32_t *range_counts, uint32_t *work_counter, int cus, hipStream_t stream, size_t min_lds, const u64 *dyn)
{
    constexpr int BLOCK = 1024;
    const uint32_t P = F1 * F2;
    size_t lds = ((size_t)P + F1) * sizeof(uint32_t);
    if (geom.chunks == 0 || geom.chunks > HJ_MAX_CHUNKS || lds > 140 * 1024) return HJGPU_EINVAL;
    // option "hist_min_lds" (diagnostics): the workgroup asks for at least this much LDS, so that nothing else fits the CU beside it
    if (min_lds > lds && min_lds <= 140 * 1024) lds = min_lds;
    static HjPerDeviceOnce once;
    if (hj_allow_dynamic_lds(reinterpret_cast<const void *>(&hist2_kernel<BLOCK>), 140 * 1024, &once) != HJGPU_OK)
        return HJGPU_EHIP;
    const uint32_t per_cu = (lds > 72 * 1024) ? 1 : 2;
    uint32_t gx = ((uint32_t)cus * per_cu + geom.chunks - 1) / geom.chunks;
    if (gx > geom.ranges_per_chunk) gx = geom.ranges_per_chunk;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(hist2_kernel<BLOCK>, dim3(gx, geom.chunks), dim3(BLOCK), lds, stream, keys, geom,
                       f1, F1, f2, F2, counts, range_counts, work_counter, dyn);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// --------------------------------------------------------------------------
// K4p: fused histogram of PRE-PARTITIONED packed tuples.  The receiving side of the multi-GPU CPRA gets its share of
// a relation pass-1-partitioned already - the exchange-level partitioning of the senders' own chunks IS pass 1
// (cpra2.cpp:1757-1827; ownership 1868-1872) - as `chunks` pieces (one per source rank), each holding this rank's
// pass-1 partitions [p1_base, p1_base + F1) in order.  What pass 2 and the join still need are the counts per
// (chunk, final partition): bin = (H(key, f1, F1tot) - p1_base) * F2 + H(key, f2, F2), one read of the tuples.
// grid = (workgroups per chunk, chunks); counts[chunk][F1 * F2] must be zeroed by the caller.
// --------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void hist_packed_kernel(const u64 *__restrict__ tuples, HjChunks ch,
                                                            uint32_t f1, uint32_t F1tot, uint32_t p1_base, uint32_t F1,
                                                            uint32_t f2, uint32_t F2, u64 *__restrict__ counts)
{
    extern __shared__ uint32_t lds_hist[];              // [F1 * F2]
    const uint32_t P = F1 * F2;
    const uint32_t chunk = blockIdx.y;
    u64 cb = 0, ce = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (q == (int)chunk) { cb = ch.b[q]; ce = ch.b[q + 1]; }
    counts += (u64)chunk * P;
    for (uint32_t i = threadIdx.x; i < P; i += BLOCK) lds_hist[i] = 0;
    __syncthreads();
    const uint4 *__restrict__ t4 = reinterpret_cast<const uint4 *>(tuples);     // two tuples per 16 bytes
    constexpr int U = 4;                                // vectors per lane in flight
    const u64 first = cb & ~1ull;                       // the 16-byte vector that holds the chunk's first tuple
    const u64 step = (u64)BLOCK * 2 * U;
    auto add = [&](uint32_t key) {
        const uint32_t p1 = hj_hash(key, f1, F1tot) - p1_base;
        const uint32_t bin = p1 * F2 + hj_hash(key, f2, F2);
        if (p1 < F1) atomicAdd(&lds_hist[bin], 1u);     // a tuple of another rank's partitions cannot be here; never index outside
    };
    // two batches in flight per lane, as in K4: the next batch's loads are issued before the current one is counted
    // (one batch at a time ran at 2.6 TB/s: every iteration waited out a full memory round trip)
    const u64 stride = (u64)gridDim.x * step;
    auto whole_at = [&](u64 base) { return base >= cb && base + step <= ce; };                  // uniform: no per-tuple predicates
    auto fetch = [&](u64 base, uint4 (&v)[U], bool whole) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u64 g = base + ((u64)u * BLOCK + threadIdx.x) * 2;
            v[u] = make_uint4(0, 0, 0, 0);
            if (whole || g < ce) v[u] = hj_load_nt(t4 + (g >> 1));                              // read once
        }
    };
    auto count = [&](u64 base, const uint4 (&v)[U], bool whole) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u64 g = base + ((u64)u * BLOCK + threadIdx.x) * 2;
            if (whole || (g >= cb && g < ce)) add(v[u].x);
            if (whole || (g + 1 >= cb && g + 1 < ce)) add(v[u].z);
        }
    };
    uint4 va[U], vb[U];
    u64 base = first + (u64)blockIdx.x * step;
    bool wa = whole_at(base), wb = false;
    if (base < ce) fetch(base, va, wa);
    while (base < ce) {
        wb = whole_at(base + stride);
        if (base + stride < ce) fetch(base + stride, vb, wb);
        count(base, va, wa);
        base += stride;
        if (base >= ce) break;
        wa = whole_at(base + stride);
        if (base + stride < ce) fetch(base + stride, va, wa);
        count(base, vb, wb);
        base += stride;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < P; i += BLOCK) {
        const uint32_t c = lds_hist[i];
        if (c) atomicAdd(&counts[i], (u64)c);
    }
}

int hj_launch_hist_packed(const u64 *tuples, const HjChunks &ch, uint32_t f1, uint32_t F1tot, uint32_t p1_base,
                          uint32_t F1, uint32_t f2, uint32_t F2, u64 *counts, int cus, hipStream_t stream)
{
    constexpr int BLOCK = 1024;
    const uint32_t P = F1 * F2;
    const size_t lds = (size_t)P * sizeof(uint32_t);
    if (ch.chunks == 0 || ch.chunks > 8 || P == 0 || lds > 140 * 1024 || p1_base + F1 > F1tot) return HJGPU_EINVAL;
    static HjPerDeviceOnce once;
    if (hj_allow_dynamic_lds(reinterpret_cast<const void *>(&hist_packed_kernel<BLOCK>), 140 * 1024, &once) != HJGPU_OK)
        return HJGPU_EHIP;
    const uint32_t per_cu = (2 * lds + 2048 <= 160 * 1024) ? 2 : 1;     // two 1024-thread workgroups per CU while both histograms fit the LDS
    uint32_t gx = ((uint32_t)cus * per_cu + ch.chunks - 1) / ch.chunks;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(hist_packed_kernel<BLOCK>, dim3(gx, ch.chunks), dim3(BLOCK), lds, stream, tuples, ch, f1, F1tot,
                       p1_base, F1, f2, F2, counts);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// Row sums of a fused histogram: out[p1] = sum over p2 of counts[p1 * F2 + p2] (hjgpu_partition_packed_counted_async: the
// pass-1 counts of a partitioning call whose histogram pass counted the receivers' second level as well).
__global__ __launch_bounds__(64) void row_sums_kernel(const u64 *__restrict__ counts, uint32_t F2, u64 *__restrict__ out)
{
    u64 s = 0;
    for (uint32_t j = threadIdx.x; j < F2; j += 64) s += counts[(u64)blockIdx.x * F2 + j];
    s = wave_reduce_sum(s);
    if (threadIdx.x == 0) hj_store(&out[blockIdx.x], s);
}

int hj_launch_row_sums(const u64 *counts, uint32_t F1, uint32_t F2, u64 *out, hipStream_t stream)
{
    hipLaunchKernelGGL(row_sums_kernel, dim3(F1), dim3(64), 0, stream, counts, F2, out);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// --------------------------------------------------------------------------
// K5b: per-range write bases of pass 1.  One workgroup per (chunk, partition):
// base[range][p] = off1[chunk][p] + sum of the counts of earlier ranges of the chunk.
// group_bins > 0 (one chunk; the grouped plans' pass 0, hjgpu_api.hip): the partitions are laid out in GROUPS of group_bins
// neighbours, dense inside a group, every group starting on a 128-byte line: group g moves up by hj_group_shift().
// own_count > 0 (one chunk; hjgpu_partition_packed_own_last_async): partitions [own_first, own_first + own_count) are
// laid out LAST, behind all others (which keep their order) - off1 stays the plain prefix of the counts.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void range_base_kernel(
    const uint32_t *__restrict__ range_counts, const u64 *__restrict__ off1,
    u64 *__restrict__ range_base, uint32_t Rc, uint32_t F1, uint32_t own_first, uint32_t own_count, uint32_t group_bins)
{
    __shared__ u64 scratch[256 / 64 + 1];
    const uint32_t c = blockIdx.x / F1, p = blockIdx.x - c * F1;
    const uint32_t per = (Rc + 255) / 256;
    const uint32_t lo = min(Rc, threadIdx.x * per), hi = min(Rc, lo + per);
    const u64 row0 = (u64)c * Rc;
    u64 sum = 0;
    for (uint32_t j = lo; j < hi; ++j) sum += range_counts[(row0 + j) * F1 + p];
    u64 first = off1[(u64)c * F1 + p];
    if (own_count) {
        const u64 ob = off1[own_first], oe = off1[own_first + own_count], n = off1[F1];
        if (p >= own_first + own_count) first -= oe - ob;
        else if (p >= own_first) first = n - (oe - ob) + (first - ob);
    }
    if (group_bins) first += hj_group_shift(off1[p / group_bins * group_bins], p / group_bins);
    u64 run = first + block_exclusive_scan<256, u64>(sum, scratch);
    for (uint32_t j = lo; j < hi; ++j) {
        hj_store(&range_base[(row0 + j) * F1 + p], run);
        run += range_counts[(row0 + j) * F1 + p];
    }
}

int hj_launch_range_base(const uint32_t *range_counts, const u64 *off1, u64 *range_base,
                         uint32_t chunks, uint32_t ranges_per_chunk, uint32_t F1, hipStream_t stream,
                         uint32_t own_first, uint32_t own_count, uint32_t group_bins)
{
    if (own_count && (chunks != 1 || (u64)own_first + own_count > F1)) return HJGPU_EINVAL;
    if (group_bins && (chunks != 1 || own_count)) return HJGPU_EINVAL;
    hipLaunchKernelGGL(range_base_kernel, dim3(chunks * F1), dim3(256), 0, stream, range_counts,
                       off1, range_base, ranges_per_chunk, F1, own_first, own_count, group_bins);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// --------------------------------------------------------------------------
// K5: plan.  One workgroup per job: block 0 = R, block 1 = S, block 2 = join
// work items.  A single workgroup scans up to 8*32768 counters; that is a few
// tens of microseconds and keeps every later kernel free of host round trips.
// --------------------------------------------------------------------------
constexpr int PLAN_BLOCK = 1024;

// Exclusive scan of f(i), i in [0, n), written to out[0..n] (out[n] = total); then(i, f(i), out[i]) is
// called for every i.  A thread's values (<= 32 for n <= 32768) are fetched together and kept in
// registers: the loads of f are global reads, and one dependent read per element and pass was most
// of these latency-bound kernels' time.
struct PlanNoop { __device__ void operator()(uint32_t, u64, u64) const {} };
template <int MAXPER = 32, typename F, typename G = PlanNoop>
__device__ void plan_scan(uint32_t n, F f, u64 *__restrict__ out, u64 base, u64 *scratch, G then = G(),
                          bool write_total = true)
{
    const uint32_t per = (n + PLAN_BLOCK - 1) / PLAN_BLOCK;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    u64 sum = 0;
    if (per <= MAXPER) {
        u64 v[MAXPER];
#pragma unroll
        for (int j = 0; j < MAXPER; ++j) { v[j] = 0; if (lo + j < hi) v[j] = f(lo + j); }
#pragma unroll
        for (int j = 0; j < MAXPER; ++j) sum += v[j];
        u64 run = base + block_exclusive_scan<PLAN_BLOCK, u64>(sum, scratch);
#pragma unroll
        for (int j = 0; j < MAXPER; ++j)
            if (lo + j < hi) { hj_store(&out[lo + j], run); then(lo + j, v[j], run); run += v[j]; }
        if (write_total && threadIdx.x == PLAN_BLOCK - 1) hj_store(&out[n], run);   // last thread's run == base + total
    } else {
        for (uint32_t i = lo; i < hi; ++i) sum += f(i);
        u64 run = base + block_exclusive_scan<PLAN_BLOCK, u64>(sum, scratch);
        for (uint32_t i = lo; i < hi; ++i) { const u64 x = f(i); hj_store(&out[i], run); then(i, x, run); run += x; }
        if (write_total && threadIdx.x == PLAN_BLOCK - 1) hj_store(&out[n], run);
    }
    __syncthreads();
}


// The plan kernels walk arrays of P <= 32768 counters with ONE workgroup.  With a blocked split (thread t owns
// partitions [t * per, (t + 1) * per)) every load and store instruction of a wave touches 64 different cache
// lines, and a single CU's address path then needs ~50 us for the ~10^5 line accesses of a plan step.  The
// two-pass plan kernels therefore walk in tiles of 2 * PLAN_BLOCK partitions: thread t handles partitions
// tile + 2t and tile + 2t + 1 (a wave covers 1 KiB of consecutive counters per instruction), one block scan per
// tile, running totals carried in registers.
// Exclusive scan of (a, b) over the workgroup, ONE barrier; totals returned; scratch = [2][2][NW] (parity-buffered:
// the slots of a call are rewritten two calls later, when every thread has passed another barrier).
__device__ __forceinline__ void plan_scan2(u64 &a, u64 &b, u64 &total_a, u64 &total_b, u64 *scratch, int parity)
{
    constexpr int NW = PLAN_BLOCK / 64;
    const int lane = hj_lane(), wave = threadIdx.x >> 6;
    const u64 ia = wave_inclusive_scan(a), ib = wave_inclusive_scan(b);
    u64 *sa = scratch + (parity * 2) * NW, *sb = sa + NW;
    if (lane == 63) { sa[wave] = ia; sb[wave] = ib; }
    __syncthreads();
    u64 before_a = 0, before_b = 0;
    total_a = 0; total_b = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const u64 xa = sa[w], xb = sb[w];
        total_a += xa; total_b += xb;
        if (w < wave) { before_a += xa; before_b += xb; }
    }
    a = ia - a + before_a;
    b = ib - b + before_b;
}

// K5, step 1: one workgroup per (chunk, relation).  A chunk's final offsets are its
// own exclusive scan plus the chunk's first row (known on the host), so the chunks
// scan in parallel.
__device__ __forceinline__ void plan_items_body(const PlanArgs &a);

template <bool PAD>
__global__ __launch_bounds__(PLAN_BLOCK) void plan_offsets_kernel(PlanArgs a)
{
    __shared__ u64 scratch[PLAN_BLOCK / 64 + 1];
    const uint32_t P = a.F1 * a.F2, C = a.chunks;
    const uint32_t c = blockIdx.x;
    // grid row 2: the join's work items (K5 step 3) ride along - they depend on the two histograms only, not on the
    // offsets, and a single-workgroup kernel of their own cost another 29 us of launch-to-launch latency per step
    if (blockIdx.y == 2) { if (c == 0) plan_items_body(a); return; }
    const int r = blockIdx.y;
    if (!((a.mask >> r) & 1u)) return;
    const u64 *__restrict__ cnt = a.counts[r] + (u64)c * P;
    u64 *off2 = a.off2[r] + (u64)c * P;
    u64 *end2 = a.end2[r] + (u64)c * P;
    // (a device-planned group: the relation's rows are known on the device only; its chunks are regular)
    const u64 n_r = a.dyn[r] ? a.dyn[r][1] : a.n[r];
    const u64 part_r = a.dyn[r] ? ((n_r / C) & ~(u64)15) : a.chunk_part[r];
    const u64 base = a.regular[r] ? part_r * c : a.chunk_beg[r][c];
    if (!PAD) {
        // dense final layout: partition q occupies [off2[q], off2[q + 1])
        plan_scan(P, [&](uint32_t i) { return cnt[i]; }, off2, base, scratch,
                  [&](uint32_t i, u64 n, u64 first) { hj_store(&end2[i], first + n); });
        // off2[P] of this chunk is the next chunk's first row (same value: chunks are contiguous)
        for (uint32_t i = threadIdx.x; i < P; i += PLAN_BLOCK) hj_store(&a.cur2[r][(u64)c * P + i], off2[i]);
        for (uint32_t p1 = threadIdx.x; p1 < a.F1; p1 += PLAN_BLOCK) {
            const u64 o = off2[(u64)p1 * a.F2];
            hj_store(&a.off1[r][(u64)c * a.F1 + p1], o);
            hj_store(&a.cur1[r][(u64)c * a.F1 + p1], o);
        }
    } else {
        // Two passes: the pass-1 output (= pass-2 input) stays dense, the FINAL layout starts every
        // partition on a 128-byte line (K6 pass 2 then claims whole lines from the front of a partition
        // and the few leftover tuples of a tile from its back).
        // The pass-1 layout is per chunk (every chunk is partitioned on its own, cpra2.cpp:1757-1827); the FINAL
        // layout is shared by all chunks: partition q is ONE line-aligned region sized for the sum of the chunks'
        // counts, and the pass-2 tiles of every chunk claim lines (front) and tail slots (back) from the same
        // cursor.  The reference reaches the same state with its memcpy gather (cpra2.cpp:1891-1959); here the
        // join then sees one piece per partition, exactly as after PHJ's passes.  Block 0 lays the final
        // partitions out (entries [0, P) of off2 / end2 / cur2), every block its own chunk's pass-1 partitions.
        __shared__ u64 scratch2[4 * (PLAN_BLOCK / 64)];
        // partition-major pass-1 layout (PlanArgs::p_major): rows of my pass-1 partitions in the chunks BEFORE mine
        __shared__ u64 before[HJGPU_MAX_FANOUT];
        const bool p_major = a.p_major != 0;
        if (p_major) {
            // one wave per pass-1 partition at a time: its lanes read the partition's F2 counters of every earlier chunk
            // (coalesced, all loads of a partition in flight together), one wave reduction per partition
            for (uint32_t p1 = threadIdx.x >> 6; p1 < a.F1; p1 += PLAN_BLOCK / 64) {
                u64 s = 0;
                for (uint32_t cc = 0; cc < c; ++cc)
                    for (uint32_t h = threadIdx.x & 63u; h < a.F2; h += 64) s += a.counts[r][(u64)cc * P + (u64)p1 * a.F2 + h];
                s = wave_reduce_sum(s);
                if ((threadIdx.x & 63u) == 0) before[p1] = s;
            }
            __syncthreads();
        }
        auto padded = [](u64 n) { return (n + HJ_LINE_TUPLES - 1) & ~(u64)(HJ_LINE_TUPLES - 1); };
        u64 drun = p_major ? (a.regular[r] ? 0 : a.chunk_beg[r][0]) : base;                // dense offset of the tile's first partition
        u64 prun = 0;                                                 // padded one (block 0; the relation starts at row 0)
        const uint32_t F2 = a.F2;
        const u64 *__restrict__ all = a.counts[r];
        const u64 *__restrict__ more = a.more[r];
        u64 *fo = a.off2[r], *fe = a.end2[r], *fc = a.cur2[r];
        auto total_of = [&](uint32_t q) -> u64 {
            u64 v[8];                                            // the first eight chunks' counters are requested before the first add
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) v[k] = k < C ? all[(u64)k * P + q] : 0;
            u64 t = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            if (more) t += more[q];                              // chunks 8 ... C - 1, summed by chunk_tail_sums_kernel
            return t;
        };
        int parity = 0;
        for (uint32_t tile = 0; tile < P; tile += 2 * PLAN_BLOCK, parity ^= 1) {
            const uint32_t q0 = tile + 2 * threadIdx.x, q1 = q0 + 1;
            const bool totals = c == 0 || p_major;
            const u64 t0 = (totals && q0 < P) ? total_of(q0) : 0, t1 = (totals && q1 < P) ? total_of(q1) : 0;
            // the dense (pass-1) layout: my chunk's rows, or - partition-major - the rows of all chunks
            const u64 n0 = p_major ? t0 : (q0 < P ? cnt[q0] : 0), n1 = p_major ? t1 : (q1 < P ? cnt[q1] : 0);
            u64 d = n0 + n1, pd = padded(t0) + padded(t1), dt, pt;
            plan_scan2(d, pd, dt, pt, scratch2, parity);
            d += drun; pd += prun;
            if (q0 < P) {
                if (q0 % F2 == 0) {
                    const u64 mine = d + (p_major ? before[q0 / F2] : 0);
                    hj_store(&a.off1[r][(u64)c * a.F1 + q0 / F2], mine); hj_store(&a.cur1[r][(u64)c * a.F1 + q0 / F2], mine);
                    if (p_major && c == 0) hj_store(&a.seg2[r][q0 / F2], d);
                }
                if (c == 0) { hj_store(&fo[q0], pd); hj_store(&fe[q0], pd + t0); hj_store(&fc[q0], (u64)0); }      // cursor: lines claimed | tail tuples << 32
            }
            if (q1 < P) {
                const u64 d1 = d + n0, p1 = pd + padded(t0);
                if (q1 % F2 == 0) {
                    const u64 mine = d1 + (p_major ? before[q1 / F2] : 0);
                    hj_store(&a.off1[r][(u64)c * a.F1 + q1 / F2], mine); hj_store(&a.cur1[r][(u64)c * a.F1 + q1 / F2], mine);
                    if (p_major && c == 0) hj_store(&a.seg2[r][q1 / F2], d1);
                }
                if (c == 0) { hj_store(&fo[q1], p1); hj_store(&fe[q1], p1 + t1); hj_store(&fc[q1], (u64)0); }
            }
            drun += dt; prun += pt;
        }
    }
    if (threadIdx.x == 0) {
        hj_store(&a.seg1[r][c], base);
        if (c == C - 1) { hj_store(&a.seg1[r][C], n_r); hj_store(&a.off1[r][(u64)C * a.F1], n_r); }
        if (PAD && a.p_major && c == 0) hj_store(&a.seg2[r][a.F1], n_r);
    }
}

// K5, step 2: tile prefixes of both passes (block r = relation r).
__global__ __launch_bounds__(PLAN_BLOCK) void plan_tiles_kernel(PlanArgs a)
{
    __shared__ u64 scratch[PLAN_BLOCK / 64 + 1];
    const uint32_t C = a.chunks;
    if (!((a.mask >> blockIdx.x) & 1u)) return;
    {
        const int r = blockIdx.x;
        // pass-1 tiles: segments are the chunks of the caller's (possibly unaligned) input
        const uint32_t al = a.in_align[r];
        const uint32_t tile1 = a.tile1, tile2 = a.tile2;
        const u64 *seg1 = a.seg1[r];
        plan_scan<8>(C, [&](uint32_t i) { return hj_tiles_of(seg1[i], seg1[i + 1], al, tile1); },
                     a.tp1[r], 0, scratch);
        // pass-2 tiles: segments are the pass-1 partitions inside the (aligned) workspace
        // (partition-major pass-1 layout: F1 segments, the chunks' regions of a partition side by side)
        const u64 *off1 = a.p_major ? a.seg2[r] : a.off1[r];
        const uint32_t F1 = a.F1, inter = a.seg_interleave;
        plan_scan<8>(a.p_major ? a.F1 : C * a.F1, [&](uint32_t i) {
                         const uint32_t sgm = inter ? (i % C) * F1 + i / C : i;
                         return hj_tiles_of(off1[sgm], off1[sgm + 1], 0, tile2);
                     }, a.tp2[r], 0, scratch);
    }
}

// K5, step 2b: per-tile descriptors of pass 2: K6 then needs ONE independent 32-byte load per tile instead of a
// search in tp2 followed by dependent reads of off1 (exposed latency on every tile).  One wave per segment, one
// lane per tile (a segment of the probe side has hundreds of tiles); 2 MB of descriptors at 64 M x 1 G, which
// one workgroup needed 20 us to write: grid of TDESC_BLOCKS workgroups per relation.
constexpr int TDESC_BLOCKS = 32;
__global__ __launch_bounds__(PLAN_BLOCK) void tile_desc_kernel(PlanArgs a)
{
    const int r = blockIdx.y;
    if (!((a.mask >> r) & 1u) || !a.tdesc[r]) return;
    const uint32_t nseg = a.p_major ? a.F1 : a.chunks * a.F1, tile2 = a.tile2;
    const u64 *__restrict__ off1 = a.p_major ? a.seg2[r] : a.off1[r];
    const u64 *__restrict__ tp2 = a.tp2[r];
    uint4 *td = a.tdesc[r];
    for (uint32_t i = blockIdx.x * (PLAN_BLOCK / 64) + (threadIdx.x >> 6); i < nseg; i += TDESC_BLOCKS * (PLAN_BLOCK / 64)) {
        const uint32_t sgm = a.seg_interleave ? (i % a.chunks) * a.F1 + i / a.chunks : i;      // the i-th segment in tile order
        const u64 gb = off1[sgm], ge = off1[sgm + 1];
        const u64 t0 = tp2[i], t1 = min(tp2[i + 1], (u64)a.tdesc_cap);
        for (u64 t = t0 + (threadIdx.x & 63); t < t1; t += 64) {
            const u64 g0 = (gb & ~3ull) + (t - t0) * tile2;
            hj_store(&td[2 * t], make_uint4((uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)ge, (uint32_t)(ge >> 32)));
            // third word: first entry of the segment's final partitions in the cursor / offset tables - shared by
            // all chunks in the line-aligned layout, per (chunk, pass-1 partition) in the dense one
            hj_store(&td[2 * t + 1], make_uint4((uint32_t)g0, (uint32_t)(g0 >> 32), (a.pad2 ? sgm % a.F1 : sgm) * a.F2, sgm));
        }
    }
}

// K5, step 3: the join's work items.  Partition q with both sides non-empty gets slices(q) = ceil(|S_q| / slice)
// probe slices times groups(q) = min(HJ_JOIN_FILL_GROUPS, ceil(|R_q| / cap)) groups of table fills: a build
// partition that fits one LDS table (the planned case) is one group.
__device__ __forceinline__ void plan_items_body(const PlanArgs &a)
{
    const uint32_t P = a.F1 * a.F2;
    const uint32_t C = a.chunks;
    const u64 *__restrict__ cr = a.counts[0];
    const u64 *__restrict__ cs = a.counts[1];
    // slice and cap are powers of two (checked by hj_launch_plan): shifts, no 64-bit divisions
    static_assert((HJ_JOIN_SLICE & (HJ_JOIN_SLICE - 1)) == 0, "slice must be a power of two");
    constexpr int SLICE_SHIFT = __builtin_ctz(HJ_JOIN_SLICE);
    const uint32_t cap = a.cap ? a.cap : 1u, cap_shift = 31u - (uint32_t)__clz((int)cap);
    // (probe slices | fill groups << 32) of a partition with nr build and ns probe rows; 0 slices = no work
    auto shape_of = [&](u64 nr, u64 ns) -> u64 {
        const u64 slices = (nr && ns) ? (ns + HJ_JOIN_SLICE - 1) >> SLICE_SHIFT : 0;
        const u64 groups = a.unique ? (u64)1 : min((u64)HJ_JOIN_FILL_GROUPS, max((u64)1, (nr + cap - 1) >> cap_shift));
        return slices | (groups << 32);
    };
    // tiles of 2 * PLAN_BLOCK partitions, thread t takes partitions tile + 2t and tile + 2t + 1 (see plan_scan2)
    __shared__ u64 scratch2[4 * (PLAN_BLOCK / 64)];
    auto rows_of = [&](const u64 *__restrict__ cnt, uint32_t q) -> u64 {
        u64 v[8];                                   // the first eight chunks: all loads are requested before the first add
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) v[c] = c < C ? cnt[(u64)c * P + q] : 0;
        u64 t = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        const u64 *__restrict__ more = cnt == cr ? a.more[0] : a.more[1];
        if (more) t += more[q];                     // chunks 8 ... C - 1 (chunk_tail_sums_kernel)
        return t;
    };
    u64 run = 0;
    int parity = 0;
    uint32_t multi = 0;                             // my partitions with work whose build rows take more than one table fill
    for (uint32_t tile = 0; tile < P; tile += 2 * PLAN_BLOCK, parity ^= 1) {
        const uint32_t q0 = tile + 2 * threadIdx.x, q1 = q0 + 1;
        const u64 r0 = q0 < P ? rows_of(cr, q0) : 0, r1 = q1 < P ? rows_of(cr, q1) : 0;
        const u64 s0 = q0 < P ? shape_of(r0, rows_of(cs, q0)) : 0;
        const u64 s1 = q1 < P ? shape_of(r1, rows_of(cs, q1)) : 0;
        multi += ((s0 & 0xFFFFFFFFull) && r0 > cap) ? 1u : 0u;
        multi += ((s1 & 0xFFFFFFFFull) && r1 > cap) ? 1u : 0u;
        const u64 n0 = (s0 & 0xFFFFFFFFull) * (s0 >> 32), n1 = (s1 & 0xFFFFFFFFull) * (s1 >> 32);
        u64 first = n0 + n1, unused = 0, total, t2;
        plan_scan2(first, unused, total, t2, scratch2, parity);
        first += run;
        if (q0 < P) {
            hj_store(&a.slice_prefix[q0], first);
            hj_store(&a.slices[q0], s0);
            // item -> partition directory: the join reads one word instead of a binary search
            for (u64 i = 0; i < n0; ++i) hj_store(&a.item_part[first + i], q0);
        }
        if (q1 < P) {
            hj_store(&a.slice_prefix[q1], first + n0);
            hj_store(&a.slices[q1], s1);
            for (u64 i = 0; i < n1; ++i) hj_store(&a.item_part[first + n0 + i], q1);
        }
        run += total;
    }
    if (threadIdx.x == 0) hj_store(&a.slice_prefix[P], run);
    // the multi-fill half of a _UNIQUE join (join_kernel<.., UNIQUE, DEDUP>) returns at once when this stays 0
    if (a.multi_fill && multi) atomicAdd(a.multi_fill, multi);
}

__global__ __launch_bounds__(PLAN_BLOCK) void plan_items_kernel(PlanArgs a) { plan_items_body(a); }

// Batch plan (see BatchPlanArgs): one workgroup per batch, thread p = pass-1 partition p (F1 <= 1024).
__global__ __launch_bounds__(PLAN_BLOCK) void batch_plan_kernel(BatchPlanArgs a)
{
    __shared__ u64 scratch[PLAN_BLOCK / 64 + 1];
    const uint32_t b = blockIdx.x, p = threadIdx.x, F1 = a.F1;
    const uint32_t rb = b * a.ranges_per_batch, re = min(a.ranges, rb + a.ranges_per_batch);
    const bool mine = p < F1;
    u64 total = 0;
    if (mine) for (uint32_t r = rb; r < re; ++r) total += a.range_counts[(u64)r * F1 + p];       // coalesced over p
    const u64 off = block_exclusive_scan<PLAN_BLOCK, u64>(total, scratch);
    u64 *boff = a.boff + (u64)b * (F1 + 1);
    if (mine) {
        hj_store(&boff[p], off);
        if (p == F1 - 1) hj_store(&boff[F1], off + total);
        u64 run = off;
        for (uint32_t r = rb; r < re; ++r) { hj_store(&a.range_base[(u64)r * F1 + p], run); run += a.range_counts[(u64)r * F1 + p]; }
    }
    __syncthreads();
    const u64 tiles = mine ? hj_tiles_of(off, off + total, 0, a.tile2) : 0;
    const u64 t0 = block_exclusive_scan<PLAN_BLOCK, u64>(tiles, scratch);
    u64 *tp = a.tp2b + (u64)b * (F1 + 1);
    if (mine) {
        hj_store(&tp[p], t0);
        if (p == F1 - 1) hj_store(&tp[F1], t0 + tiles);
        uint4 *td = a.tdesc + (u64)b * a.tdesc_cap * 2;
        const u64 gb = off, ge = off + total;
        for (u64 t = t0; t < t0 + tiles && t < a.tdesc_cap; ++t) {
            const u64 g0 = (gb & ~3ull) + (t - t0) * a.tile2;
            hj_store(&td[2 * t], make_uint4((uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)ge, (uint32_t)(ge >> 32)));
            hj_store(&td[2 * t + 1], make_uint4((uint32_t)g0, (uint32_t)(g0 >> 32), p * a.F2, p));
        }
    }
}

int hj_launch_batch_plan(const BatchPlanArgs &a, uint32_t batches, hipStream_t stream)
{
    if (a.F1 == 0 || a.F1 > (uint32_t)PLAN_BLOCK || batches == 0) return HJGPU_EINVAL;
    hipLaunchKernelGGL(batch_plan_kernel, dim3(batches), dim3(PLAN_BLOCK), 0, stream, a);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// More than 8 chunks (the reference's larger thread counts): more[q] = the counters of chunks 8 ... C - 1 added up, so that the plan
// kernels keep their eight loads per partition (a loop over the chunks there cost plan_offsets_kernel 17 VGPRs of spills).
__global__ __launch_bounds__(256) void chunk_tail_sums_kernel(const u64 *__restrict__ counts, uint32_t C, uint32_t P, u64 *__restrict__ more)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= P) return;
    u64 t = 0;
    for (uint32_t c = 8; c < C; ++c) t += counts[(u64)c * P + q];
    hj_store(&more[q], t);
}

int hj_launch_plan(const PlanArgs &a, hipStream_t stream)
{
    const uint32_t P_all = a.F1 * a.F2;
    for (int r = 0; r < 2; ++r) {
        if (a.chunks > 8 && !a.more[r]) return HJGPU_EINVAL;
        // (bit 2, the join's work items, reads both relations' sums: a probe of a prepared build side has the build side's from its build)
        if (a.more[r] && ((a.mask >> r) & 1u))
            hipLaunchKernelGGL(chunk_tail_sums_kernel, dim3((P_all + 255) / 256), dim3(256), 0, stream, a.counts[r], a.chunks, P_all, a.more[r]);
    }
    if (a.mask & 3u) {
        const unsigned rows = (a.mask & 4u) ? 3u : 2u;          // row 2 = the join's work items
        if ((a.mask & 4u) && ((a.cap & (a.cap - 1)) || a.slice != (uint32_t)HJ_JOIN_SLICE)) return HJGPU_EINVAL;
        if (a.pad2) hipLaunchKernelGGL(plan_offsets_kernel<true>, dim3(a.chunks, rows), dim3(PLAN_BLOCK), 0, stream, a);
        else hipLaunchKernelGGL(plan_offsets_kernel<false>, dim3(a.chunks, rows), dim3(PLAN_BLOCK), 0, stream, a);
    }
    if (a.mask & 3u) hipLaunchKernelGGL(plan_tiles_kernel, dim3(2), dim3(PLAN_BLOCK), 0, stream, a);
    if ((a.mask & 3u) && (a.tdesc[0] || a.tdesc[1]))
        hipLaunchKernelGGL(tile_desc_kernel, dim3(TDESC_BLOCKS, 2), dim3(PLAN_BLOCK), 0, stream, a);
    if ((a.mask & 4u) && ((a.cap & (a.cap - 1)) || a.slice != (uint32_t)HJ_JOIN_SLICE)) return HJGPU_EINVAL;
    if ((a.mask & 4u) && !(a.mask & 3u)) hipLaunchKernelGGL(plan_items_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, a);   // a probe of a prepared build side
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// Grouped plans on the device (hj_internal.hpp: hj_launch_group_desc): one thread per group.
__global__ __launch_bounds__(256) void group_desc_kernel(const u64 *__restrict__ roff, const u64 *__restrict__ soff, uint32_t G, uint32_t bins,
                                                         u64 cap_r, u64 cap_s, u64 n_r, u64 n_s, u64 *__restrict__ desc, uint32_t *skew)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const u64 rb = roff[(u64)g * bins], re = roff[(u64)(g + 1) * bins];
    const u64 sb = soff[(u64)g * bins], se = soff[(u64)(g + 1) * bins];
    u64 rn = re - rb, sn = se - sb;
    // beyond what the workspace was planned for - or offsets that are no prefix of the relations' rows (a caller who overlaps joins of ONE
    // context on several streams races on its workspace: never an address outside the columns): skipped, the join flagged
    if (rn > cap_r || sn > cap_s || re < rb || se < sb || re > n_r || se > n_s) { atomicOr(skew, 1u); rn = 0; sn = 0; }
    if (rn == 0 || sn == 0) { rn = 0; sn = 0; }                                 // nothing can match
    hj_store(&desc[4 * (u64)g + 0], rb + hj_group_shift(rb, g)); hj_store(&desc[4 * (u64)g + 1], rn);
    hj_store(&desc[4 * (u64)g + 2], sb + hj_group_shift(sb, g)); hj_store(&desc[4 * (u64)g + 3], sn);
}

int hj_launch_group_desc(const u64 *roff, const u64 *soff, uint32_t G, uint32_t bins, u64 cap_r, u64 cap_s, u64 n_r, u64 n_s, u64 *desc, uint32_t *skew,
                         hipStream_t stream)
{
    if (!G || !bins) return HJGPU_EINVAL;
    hipLaunchKernelGGL(group_desc_kernel, dim3((G + 255) / 256), dim3(256), 0, stream, roff, soff, G, bins, cap_r, cap_s, n_r, n_s, desc, skew);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

__global__ void group_result_kernel(const u64 *__restrict__ state, const uint32_t *__restrict__ skew, u64 *__restrict__ d_result)
{
    if (threadIdx.x < 4) hj_store(&d_result[threadIdx.x], *skew ? ~(u64)0 : state[threadIdx.x]);
}

int hj_launch_group_result(const hjgpu_result *state, const uint32_t *skew, hjgpu_result *d_result, hipStream_t stream)
{
    hipLaunchKernelGGL(group_result_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const u64 *>(state), skew, reinterpret_cast<u64 *>(d_result));
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// Plain exclusive scan of n (<= 2^20) uint64 counters: hjgpu_partition's offsets.
__global__ __launch_bounds__(PLAN_BLOCK) void exscan_kernel(const u64 *__restrict__ in,
                                                            u64 *__restrict__ out, uint32_t n)
{
    __shared__ u64 scratch[PLAN_BLOCK / 64 + 1];
    plan_scan(n, [&](uint32_t i) { return in[i]; }, out, 0, scratch);
}

int hj_launch_exscan(const u64 *in, u64 *out, uint32_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, in, out, n);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// counts[p] = off[p+1] - off[p]: turns caller-provided partition offsets
// (hjgpu_join_partitions) back into the histogram form the plan kernel reads.
__global__ void offsets_to_counts_kernel(const u64 *__restrict__ off, u64 *__restrict__ counts, uint32_t P)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) hj_store(&counts[i], off[i + 1] - off[i]);
}

int hj_launch_offsets_to_counts(const u64 *off, u64 *counts, uint32_t P, hipStream_t stream)
{
    hipLaunchKernelGGL(offsets_to_counts_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, off, counts, P);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// --------------------------------------------------------------------------
// K6: scatter one pass.
//   RANGED (pass 1): a workgroup walks whole ranges of tiles; the output position
//     of every (range, partition) was computed by K5b, so the bin-owner thread
//     keeps the write cursor of its bins in registers - no global atomics (with
//     one shared cursor per partition, 512 workgroups hammering the same few
//     cache lines cost 6.5 of 9.7 ms at |S| = 1G).
//   !RANGED (pass 2): a workgroup owns a contiguous run of tiles, i.e. stays in
//     the same pass-1 partition for a while, so each partition's F2 cursors are
//     shared by only a few workgroups; one returning atomic per (tile, partition).
// --------------------------------------------------------------------------
// Column formats: the caller's relations are separate key / payload columns
// (hj.h:1-72).  Between the passes and into the join the tuples travel PACKED
// (payload << 32 | key, 8 bytes): a run of L tuples is then one 8L-byte burst
// instead of two 4L-byte bursts in two arrays, which is what the DRAM sees.
// IN_PACKED / OUT_PACKED select the format on either side.
//
// CARRY (pass 1 of the join pipeline: private cursors + packed output): every 128-byte
// line of the output leaves the CU in one piece.  A run that ends in the middle of a line
// is cut at the line boundary and its tail (< 16 tuples) waits in an LDS carry buffer for the
// next tile of the range; the lanes that write the head of that tile's run also write the
// carried tuples in front of it, in the same loop iteration.
// Measured with tools/ubench_scatter_align.hip (same traffic, no sort), ms per 1G tuples:
//   runs that start and end on 128-byte lines      3.3-3.4 at fan-out 16, 128 and 512
//                                                  (even ONE line per partition and tile)
//   the same runs shifted by 40 bytes              3.5 / 4.0 / 5.1
//   a line completed one barrier later (split)     3.7 (128) / 5.8 (512)
//   a line written by two waves in the same sweep  3.3 (no cost)
// i.e. a partial line is not kept in the XCD's L2 until somebody completes it: it becomes a
// read-modify-write at the memory side unless the rest arrives at about the same time.
// NTP: the 8-byte (partial-line) stores are non-temporal too - every launch that is not solo (k6_store8).  A template parameter:
// as a run-time flag around each store the two branches were merged by the compiler into ONE plain store (the nt hint does not
// survive the merge; found in the ISA after a run in which "non-temporal" partial stores cost nothing).
template <int BLOCK, int VPT, bool RANGED, bool IN_PACKED, bool OUT_PACKED, bool CARRY, bool NTP = true>
__global__ __launch_bounds__(BLOCK) void scatter_kernel(ScatterArgs a)
{
    static_assert(!CARRY || (RANGED && OUT_PACKED), "carry needs private cursors and packed output");
    constexpr int TILE = BLOCK * VPT * 4;
    constexpr int NW = BLOCK / 64;
    constexpr int BPT = (1024 + BLOCK - 1) / BLOCK;                 // max bins per thread (F <= 1024)
    constexpr uint32_t LINE = HJ_LINE_TUPLES;                       // packed tuples per 128-byte line
    constexpr uint32_t UNIT = HJ_STREAM_UNIT;                       // output slots a 16-lane group moves at a time
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t F = a.F;
    const uint32_t Fpad = (F + 3) & ~3u;
    u64 *delta = reinterpret_cast<u64 *>(smem);                     // [Fpad]  packed out: output position of the run; else output - local position
    u64 *stage = delta + Fpad;                                      // [TILE]  payload << 32 | key, sorted by partition
    u64 *carry = stage + TILE;                                      // CARRY: [Fpad][LINE] tails waiting for their line
    u64 *tinfo = carry;                                             // pass 2: [Fpad] (position of the run's tail << 4) | tail tuples
    uint32_t *hist = reinterpret_cast<uint32_t *>(carry + (CARRY ? Fpad * LINE : (RANGED ? 0 : Fpad)));   // [Fpad]  counts, then local bases
    uint32_t *meta = hist + Fpad;                                   // [Fpad] tuples leaving this tile | carried ones among them << 16
    uint32_t *left = meta + Fpad;                                   // CARRY: [Fpad] first staying index | carry offset << 16 | count << 20
    uint32_t *wsum = left + (CARRY ? Fpad : 0);                     // [NW + 6]; [NW + 1] = number of runs longer than one unit, [NW + 4..5] = tickets,
                                                                    // [NW + 2] = this tile's heavy partition (count << 10 | bin, 0 = none),
                                                                    // [NW + 3] = the same for the next tile
    uint32_t *heavy = wsum + NW + 6;                                // [HJ_MAX_HEAVY] partitions whose run is longer than one unit

    const int tid = threadIdx.x;
    // pass 1 of a device-planned group (ScatterArgs::dyn): first row and rows of the input come from device memory (uniform: scalar loads)
    Pass1Geom geom = a.geom;
    u64 in_row0 = 0;
    if (RANGED && !IN_PACKED && a.dyn) { in_row0 = a.dyn[0]; geom.n = a.dyn[1]; geom.part = (geom.n / geom.chunks) & ~(u64)15; }
    // IN_PACKED inputs are workspace arrays (in_align == 0): tuple g lives in uint4 g/2
    const uint4 *__restrict__ k4 = reinterpret_cast<const uint4 *>(IN_PACKED ? a.kin : a.kin + in_row0 - a.in_align);
    const uint4 *__restrict__ v4 = reinterpret_cast<const uint4 *>(IN_PACKED ? a.kin : a.vin + in_row0 - a.in_align);
    const uint32_t factor = a.factor;
    const uint32_t bpt = (F + BLOCK - 1) / BLOCK;                   // bins per thread in the scan
    u64 mycur[BPT];                                                 // RANGED: cursors of my bins
    uint32_t mycc[BPT];                                             // CARRY: tuples of my bins waiting in `carry`
#pragma unroll
    for (int i = 0; i < BPT; ++i) { mycur[i] = 0; mycc[i] = 0; }
#ifdef HJ_FORENSIC_PRIVATE_WORD
    // forensic builds only (tools/build_variant.py <name> -DHJ_FORENSIC_PRIVATE_WORD=1, never the product): pass 1 carries one private
    // word, written once and never read - round 4's "variant 9", with which K6 loses stores in a third of the steps beside another
    // stream's kernels: the high-rate form of the loss, used to tell a lost store from a stale read (tools/scratch_two_streams.py --recheck)
    volatile uint32_t forensic_private[2];
    if (RANGED) { forensic_private[0] = 0; forensic_private[1] = (uint32_t)threadIdx.x; }
#endif

    // ---- the sequence of tiles this workgroup processes ------------------------------
    // Work is CLAIMED, not owned: pass 1 takes whole ranges, pass 2 single tiles, in order, from a
    // global ticket counter (zeroed per launch).  A static split (range r = blockIdx + k * grid)
    // makes the kernel as slow as its unluckiest workgroup: when other kernels hold some CUs - the
    // RCCL transfer of the build side on the multi-GPU path - the workgroups that found no CU
    // start only after the others have finished their share.  With tickets whoever runs, works.
    // Tickets are prefetched: thread 0 issues the atomic for ticket n+2 when ticket n+1 is taken
    // and deposits it in LDS a few barriers later, so nobody ever waits for an atomic.
    struct Tile { u64 gb, ge, g0, cursor_row; uint32_t range; bool new_range, last_in_range, empty, valid; };
    uint32_t *claim = wsum + NW + 4;                                 // [2] prefetched tickets
    int claim_parity = 0;
    uint32_t pending_ticket = 0;                                    // thread 0: ticket on its way
    int pending_slot = -1;                                          // thread 0: where it goes (-1: none)
    if (tid == 0) {
        const uint32_t t0 = atomicAdd(a.work_counter, 2u);
        claim[0] = t0; claim[1] = t0 + 1;
    }
    __syncthreads();
    // H(key, f, 2^k) = (key * f) >> (32 - k): the same function, one multiply less
    const bool f_pow2 = F > 1 && (F & (F - 1)) == 0;
    const uint32_t f_shift = f_pow2 ? 32u - (uint32_t)__builtin_ctz(F) : 0u;
    auto part_of = [&](uint32_t key) -> uint32_t {
        const uint32_t x = key * factor;
        return f_pow2 ? x >> f_shift : __umulhi(x, F);
    };
    // A lane-dependent zero the compiler cannot see through: with a provably uniform address LLVM's atomic
    // optimizer aggregates the add over the wave and needs its result at once (s_waitcnt vmcnt(0) +
    // readfirstlane right after the atomic), which stalls wave 0 for the round trip on every tile of pass 2.
    // It is produced where it is used, every time (volatile): computed once in front of the tile loop, the compiler
    // kept `work_counter + zero` - and, the same way, `part_start + tid` / `part_end + tid` below - as per-thread 64-bit
    // addresses across the whole loop, three register pairs that pass 2 at the 128-VGPR cap had to spill to SCRATCH.
    auto opaque_zero_now = []() -> uint32_t {
        uint32_t z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        return z;
    };
    auto take_ticket = [&]() -> uint32_t {                          // uniform; at most once between two deposits
        const uint32_t t = claim[claim_parity];
        if (tid == 0) { pending_ticket = atomicAdd(a.work_counter + opaque_zero_now(), 1u); pending_slot = claim_parity; }
        claim_parity ^= 1;
        return t;
    };
    auto deposit_ticket = [&]() {                                   // >= 1 barrier after the take, >= 1 before the slot's next take
        if (tid == 0 && pending_slot >= 0) { claim[pending_slot] = pending_ticket; pending_slot = -1; }
    };
    // RANGED state
    uint32_t r_cur = 0;
    u64 rt = 0, rt_end = 0, r_gb = 0, r_ge = 0;
    bool r_open = false, exhausted = false;
    const u64 total_tiles = RANGED ? 0 : a.tile_prefix[a.nseg];
    auto next_tile = [&]() -> Tile {
        Tile t;
        t.valid = false; t.new_range = false; t.last_in_range = false; t.empty = false; t.range = 0;
        t.gb = t.ge = t.g0 = t.cursor_row = 0;
        if (exhausted) return t;
        if (RANGED) {
            const uint32_t Rc = geom.ranges_per_chunk;
            const uint32_t nranges = a.range_count ? a.range_count : Rc * geom.chunks;
            if (!r_open || rt >= rt_end) {
                r_cur = take_ticket();
                if (r_cur >= nranges) { exhausted = true; return t; }
                r_cur += a.range_begin;
                const uint32_t c = r_cur / Rc, j = r_cur - c * Rc;
                const u64 cb = geom.beg(c), ce = geom.beg(c + 1);
                r_gb = geom.align + cb; r_ge = geom.align + ce;
                const u64 tiles = hj_tiles_of(cb, ce, geom.align, TILE);
                rt = tiles * j / Rc; rt_end = tiles * (j + 1) / Rc;
                r_open = true;
                t.new_range = true;
                if (rt >= rt_end) {                                 // a range without tiles (small inputs): a tile without tuples
                    t.range = r_cur; t.valid = true; t.empty = true; t.last_in_range = true;
                    return t;
                }
            }
            t.gb = r_gb; t.ge = r_ge; t.g0 = (r_gb & ~3ull) + rt * (u64)TILE;
            t.range = r_cur; t.valid = true;
            ++rt;
            t.last_in_range = rt >= rt_end;                         // CARRY: everything still waiting goes out
        } else {
            const u64 t_cur = take_ticket();
            if (t_cur >= total_tiles) { exhausted = true; return t; }
            // one independent 32-byte read of the descriptor K5 wrote for this tile (in_align == 0)
            const uint4 d0 = a.tile_desc[2 * t_cur], d1 = a.tile_desc[2 * t_cur + 1];
            t.gb = (u64)d0.x | ((u64)d0.y << 32);
            t.ge = (u64)d0.z | ((u64)d0.w << 32);
            t.g0 = (u64)d1.x | ((u64)d1.y << 32);
            t.cursor_row = d1.z;
            t.valid = true;
        }
        return t;
    };

    // The loaded vectors themselves are the loop-carried state (no per-element copies,
    // no zero-initialised phi): the loads of tile t+1 are issued unconditionally - the
    // vector index is clamped into the segment, validity is decided later from the
    // coordinates - so the compiler has no reason to wait for them before the back edge.
    uint4 kq[VPT], vq[VPT];
    auto load_tile = [&](const Tile &t) {
        if (t.empty) return;                              // no tuples: the registers keep whatever they hold
        const u64 g_last = (t.ge - 1) & ~3ull;            // last vector that overlaps the segment
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            u64 g = t.g0 + (u64)(j * BLOCK + tid) * 4;
            g = g < g_last ? g : g_last;
            if (IN_PACKED) { kq[j] = k4[g >> 1]; vq[j] = k4[(g >> 1) + 1]; }    // 4 tuples = 2 x 16 bytes
            else { kq[j] = k4[g >> 2]; vq[j] = v4[g >> 2]; }
        }
    };
    // tuple c (0..3) of vector j
    auto key_of = [&](int j, int c) -> uint32_t {
        if (IN_PACKED) return c == 0 ? kq[j].x : c == 1 ? kq[j].z : c == 2 ? vq[j].x : vq[j].z;
        return c == 0 ? kq[j].x : c == 1 ? kq[j].y : c == 2 ? kq[j].z : kq[j].w;
    };
    auto val_of = [&](int j, int c) -> uint32_t {
        if (IN_PACKED) return c == 0 ? kq[j].y : c == 1 ? kq[j].w : c == 2 ? vq[j].y : vq[j].w;
        return c == 0 ? vq[j].x : c == 1 ? vq[j].y : c == 2 ? vq[j].z : vq[j].w;
    };

    Tile cur = next_tile();
    deposit_ticket();                                               // the loop's first take needs its slot refilled
    __syncthreads();
    if (!cur.valid) return;
    load_tile(cur);
    bool have_left = false;                                         // CARRY: the previous tile may have left tails in `stage`
    bool have_left_or_prev = false;                                 // a previous tile exists (its skew verdict is in wsum[NW + 3])
    // diagnostics: thread 0 adds the s_memtime ticks between consecutive barriers to prof[phase]
    u64 t_prev = (a.prof && tid == 0) ? __builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](int phase) {
        if (a.prof && tid == 0) {
            const u64 now = __builtin_amdgcn_s_memtime();
            atomicAdd(&a.prof[phase], now - t_prev);
            t_prev = now;
        }
    };
    // A run longer than one stream-out unit gets its further units listed (partition | unit << 16): the
    // stream-out deals the partitions' first units and these round-robin to the 16-lane groups.
    auto add_units = [&](uint32_t bin, uint32_t slots) {
        if (slots > UNIT) {
            const uint32_t extra = (slots - 1) / UNIT;
            const uint32_t at = atomicAdd(&wsum[NW + 1], extra);
            for (uint32_t u = 0; u < extra; ++u) heavy[at + u] = bin | ((u + 1) << 16);
        }
    };
    for (;;) {
        if (RANGED && cur.new_range) {
#pragma unroll
            for (int i = 0; i < BPT; ++i) {
                const uint32_t bin = tid * bpt + i;
                if ((uint32_t)i < bpt && bin < F) mycur[i] = a.range_base[(u64)cur.range * F + bin];
                mycc[i] = 0;                                        // the previous range flushed its carry
            }
        }
        if (CARRY && have_left) {
            // tails of the previous tile's runs (still in `stage`, described by `left`) move
            // behind whatever already waits in the carry; nobody touches either until the sort
            for (uint32_t idx = tid; idx < F * LINE; idx += BLOCK) {
                const uint32_t p = idx / LINE, j = idx % LINE;
                const uint32_t m = left[p];
                if (j < (m >> 20)) carry[p * LINE + ((m >> 16) & 0xFu) + j] = stage[(m & 0xFFFFu) + j];
            }
        }
        for (uint32_t i = tid; i < F; i += BLOCK) hist[i] = 0;
        if (tid == 0) { wsum[NW + 1] = 0; wsum[NW + 2] = have_left_or_prev ? wsum[NW + 3] : 0; wsum[NW + 3] = 0; }
        hj_barrier_lds();
        stamp(0);
        // the next tile's descriptor (pass 2: one read of K5's table; a scalar load, which the next
        // barrier's lgkmcnt(0) would expose) is requested here so that it lands during the ranking
        const Tile nxt = next_tile();

        // ---- rank every tuple inside its partition (ds_add_rtn_u32) --------------
        // Skew: when one partition takes a large share of the tile (a heavy-hitter key, e.g. a Zipf
        // probe side) many lanes of an instruction add to the SAME LDS word and the adds serialise.
        // The previous tile names its largest partition when that held > 1/16 of the tile; its lanes
        // are then served by ONE add of the group's size, with ranks from mbcnt.
        const uint32_t hot_word = wsum[NW + 2];
        const bool hot = hot_word != 0;
        const uint32_t hot_p = hot_word & 1023u;
        uint32_t pr[VPT * 4];
        uint32_t hot_n = 0;                                             // heavy-partition tuples of this wave
        // interior tiles (all but the first and last of a segment) need no per-tuple bounds checks
        const bool interior = cur.g0 >= cur.gb && cur.g0 + (u64)TILE <= cur.ge;
        // Interior tiles without a hot partition (all but the first / last tile of a segment, uniform keys) take a
        // straight-line path: no validity predicates, i.e. no exec-mask bookkeeping around every LDS atomic (the
        // scalar unit issues ~750 instructions per lane and tile in the general path, one scalar unit per CU).
        if (interior && !hot) {
            // (the power-of-two test is hoisted: as a per-tuple select it became a scalar branch per tuple)
            auto rank_all = [&](auto pow2) {
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t x = key_of(j, c) * factor;
                        const uint32_t p = decltype(pow2)::value ? x >> f_shift : __umulhi(x, F);
                        pr[j * 4 + c] = (p << 16) | atomicAdd(&hist[p], 1u);
                    }
                }
            };
            if (f_pow2) rank_all(std::true_type()); else rank_all(std::false_type());
        } else {
            auto partition_ids = [&](auto checked) {
    #pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const u64 g = cur.g0 + (u64)(j * BLOCK + tid) * 4;
    #pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool valid = !decltype(checked)::value || ((g + c >= cur.gb) && (g + c < cur.ge));
                        const uint32_t p = valid ? part_of(key_of(j, c)) : 0xFFFFFFFFu;
                        pr[j * 4 + c] = p;
                        if (hot) hot_n += (uint32_t)__popcll(__ballot(p == hot_p));
                    }
                }
            };
            if (interior) partition_ids(std::false_type()); else partition_ids(std::true_type());
            uint32_t hot_base = 0;
            if (hot && hot_n) {                                             // ONE add per wave for all its heavy tuples
                if (hj_lane() == 0) hot_base = atomicAdd(&hist[hot_p], hot_n);
                hot_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)hot_base);
            }
    #pragma unroll
            for (int k = 0; k < VPT * 4; ++k) {
                const uint32_t p = pr[k];
                uint32_t code = 0xFFFFFFFFu;
                if (hot) {
                    const u64 grp = __ballot(p == hot_p);
                    if (p == hot_p)
                        code = (p << 16) | (hot_base + __builtin_amdgcn_mbcnt_hi((uint32_t)(grp >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)grp, 0u)));
                    hot_base += (uint32_t)__popcll(grp);
                }
                if (p != 0xFFFFFFFFu && !(hot && p == hot_p)) code = (p << 16) | atomicAdd(&hist[p], 1u);
                pr[k] = code;
            }
        }
        hj_barrier_lds();
        stamp(1);

        // ---- local bases + one output run per non-empty partition --------------
        // Pass 2 claims its runs with returning global atomics; they are ISSUED here, as soon
        // as the counts exist, and CONSUMED only after the LDS sort, so their round trip
        // (1-2 us under load) overlaps the scan and the staging instead of stalling the tile.
        uint32_t cnt[BPT], lb[BPT];
        u64 dst[BPT];
        uint32_t carried[BPT], emitted[BPT], coff[BPT];                // CARRY only
        u64 pstart[BPT], pend[BPT];                                    // pass 2, aligned claims: my partitions' bounds
#pragma unroll
        for (int i = 0; i < BPT; ++i) { pstart[i] = 0; pend[i] = 0; }
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const uint32_t bin = tid * bpt + i;
            const bool mine = (uint32_t)i < bpt && bin < F;
            cnt[i] = mine ? hist[bin] : 0;
            sum += cnt[i];
            dst[i] = 0; carried[i] = emitted[i] = coff[i] = 0;
            if (CARRY) {
                if (mine) {
                    // c tuples wait in the carry, cnt[i] are new: emit up to the last line boundary
                    // that [cur0, cur0 + c + cnt) reaches (everything on the range's last tile)
                    const uint32_t c = mycc[i], avail = c + cnt[i];
                    const u64 cur0 = mycur[i];
                    uint32_t e = avail;
                    if (!cur.last_in_range) {
                        const u64 end = (cur0 + avail) & ~(u64)(LINE - 1);
                        e = end > cur0 ? (uint32_t)(end - cur0) : 0u;
                    }
                    if (e) carried[i] = c;                              // e >= c: cur0 + c lies before the boundary
                    else coff[i] = c;                                   // nothing leaves: the new tuples join the carry
                    emitted[i] = e;
                    dst[i] = cur0;
                    mycur[i] = cur0 + e; mycc[i] = avail - e;           // < LINE, and 0 after the range's last tile
                }
            } else if (cnt[i]) {
                if (RANGED) { dst[i] = mycur[i]; mycur[i] = dst[i] + cnt[i]; }
                else if (a.aligned_claims) {
                    // whole lines from the front of the partition, the last (< 16) tuples from its back:
                    // one returning atomic carries both claims (lines | tail tuples << 32)
                    const uint32_t front = cnt[i] & ~(LINE - 1), tail = cnt[i] & (LINE - 1);
                    // (the index is laundered per tile: no per-thread address lives across the tile loop, see opaque_zero_now)
                    uint32_t at = cur.cursor_row + bin;
                    asm volatile("" : "+v"(at));                    // an index the compiler cannot take apart
                    dst[i] = atomicAdd(&a.cursors[at], ((u64)tail << 32) | (u64)(front / LINE));
                    pstart[i] = a.part_start[at];
                    pend[i] = a.part_end[at];
                } else {
                    uint32_t at = cur.cursor_row + bin;
                    asm volatile("" : "+v"(at));                    // an index the compiler cannot take apart (see opaque_zero_now)
                    dst[i] = atomicAdd(&a.cursors[at], (u64)cnt[i]);
                }
            }
        }
        uint32_t biggest = 0;                                           // largest of my bins: count << 10 | bin
#pragma unroll
        for (int i = 0; i < BPT; ++i)
            if (cnt[i] > (uint32_t)TILE / 16) biggest = max(biggest, (cnt[i] << 10) | (tid * bpt + i));
        uint32_t run = block_exclusive_scan<BLOCK, uint32_t, true>(sum, wsum);
        if (biggest) atomicMax(&wsum[NW + 3], biggest);                 // rare; nobody reads it before the next tile
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const uint32_t bin = tid * bpt + i;
            lb[i] = run;
            if ((uint32_t)i < bpt && bin < F) {
                hist[bin] = run;
                if (CARRY) {
                    const uint32_t fresh = emitted[i] - carried[i];     // new tuples that leave with this tile
                    meta[bin] = emitted[i] | (carried[i] << 16);
                    left[bin] = (run + fresh) | (coff[i] << 16) | ((cnt[i] - fresh) << 20);
                    delta[bin] = dst[i];
                    add_units(bin, emitted[i] + ((uint32_t)dst[i] & (LINE - 1)));
                } else if (OUT_PACKED) meta[bin] = (!RANGED && a.aligned_claims) ? (cnt[i] & ~(LINE - 1)) : cnt[i];
            }
            run += cnt[i];
        }
        hj_barrier_lds();
        stamp(2);
        const uint32_t tile_count = wsum[NW];

        // ---- counting sort inside LDS --------------------------------------------
        if (interior) {                                                // every tuple of the tile is valid
            // all partition bases of a group are requested before the first is used (one LDS round trip per group, not one
            // per tuple).  Pass 2 at 4 vectors per thread takes two groups of 8: with all 16 bases live next to the 16 ranks
            // and the 32 registers of the next tile's loads the kernel needed 134 VGPRs, i.e. 6 spilled to SCRATCH at the
            // 128-register cap of a 1024-thread workgroup - and a K6 launch with a private segment loses stores when
            // kernels of another stream run beside it (DESIGN section 3 "Round 3 / 4", profiles/r04_scratch_repro.txt: the
            // private values themselves are never wrong; same machine code with and without the descriptor bit).
            // tests/test_kernel_resources.py keeps every shipped instance free of scratch.
            constexpr int GROUP = (!RANGED && VPT == 4) ? 8 : VPT * 4;
#pragma unroll
            for (int k0 = 0; k0 < VPT * 4; k0 += GROUP) {
                uint32_t pbase[GROUP];
#pragma unroll
                for (int k = 0; k < GROUP; ++k) pbase[k] = hist[pr[k0 + k] >> 16];
#pragma unroll
                for (int k = 0; k < GROUP; ++k) {
                    const int j = (k0 + k) / 4, c = (k0 + k) % 4;
                    const uint32_t pos = pbase[k] + (pr[k0 + k] & 0xFFFFu);
                    stage[pos] = (u64)key_of(j, c) | ((u64)val_of(j, c) << 32);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t code = pr[j * 4 + c];
                    if (code != 0xFFFFFFFFu) {
                        const uint32_t pos = hist[code >> 16] + (code & 0xFFFFu);
                        stage[pos] = (u64)key_of(j, c) | ((u64)val_of(j, c) << 32);
                    }
                }
            }
        }
        // consume the claims (pass 2: the atomics have had the scan and the sort to return) ...
        if (!CARRY) {
#pragma unroll
            for (int i = 0; i < BPT; ++i)
                if (!RANGED && OUT_PACKED && a.aligned_claims) {
                    const uint32_t bin = tid * bpt + i;
                    if ((uint32_t)i < bpt && bin < F) {
                        const uint32_t front = cnt[i] & ~(LINE - 1), tail = cnt[i] & (LINE - 1);
                        delta[bin] = pstart[i] + (u64)LINE * (uint32_t)dst[i];               // lines claimed before mine
                        tinfo[bin] = ((pend[i] - (dst[i] >> 32) - tail) << 4) | tail;        // tails fill the back, downwards
                        add_units(bin, front);
                    }
                } else if (cnt[i]) {
                    if (OUT_PACKED) {
                        delta[tid * bpt + i] = dst[i];
                        add_units(tid * bpt + i, cnt[i] + ((uint32_t)dst[i] & (LINE - 1)));
                    } else delta[tid * bpt + i] = dst[i] - lb[i];
                }
        }
        // ... and only then start the next tile's loads: the vector-memory counter is in
        // order, so any wait on an older result placed after these loads would also wait for
        // them.  Nothing below touches them until the next tile is ranked, and the barriers are
        // LDS-only, so they stay in flight during the whole stream-out.
        deposit_ticket();
        if (nxt.valid) load_tile(nxt);
        hj_barrier_lds();
        stamp(3);

        if (OUT_PACKED) {
            // ---- stream out, one 16-lane group per run --------------------------------
            // A group walks a partition's run in 128-byte lines of the OUTPUT: lane `sub` moves the two
            // tuples of its 16-byte slot, so a line leaves the CU as one piece of one instruction,
            // no hash and no per-tuple lookups are needed, and a store carries 16 bytes per lane.
            // A unit is (partition, UNIT consecutive output slots); units are dealt round-robin.
            // A unit is (partition, UNIT consecutive output slots): every partition's first unit plus the
            // listed further units of the runs longer than that (small fan-outs; a heavy-hitter key under
            // skew: up to the whole tile in one partition), dealt round-robin to the groups.
            constexpr uint32_t NG = BLOCK / 16;
            const uint32_t gid = tid >> 4, sub = tid & 15;
            u64 *__restrict__ out64 = reinterpret_cast<u64 *>(a.kout);
            auto move_unit = [&](uint32_t p, uint32_t c) {
                const uint32_t m = meta[p], e = m & 0xFFFFu, fc = m >> 16;
                if (e == 0) return;
                const u64 d0 = delta[p];
                const uint32_t off = (uint32_t)d0 & (LINE - 1);     // slot of the run's first tuple inside its line
                const uint32_t lim = off + e, s_end = min(lim, (c + 1) * UNIT);
                const u64 base = d0 - off;
                const uint32_t src0 = hist[p];                      // sorted index of the first fresh tuple
                // ONE LDS read per tuple: the address is selected, not the value
                auto fetch = [&](uint32_t k) -> u64 { const u64 *src = k < fc ? &carry[p * LINE + k] : &stage[src0 + (k - fc)]; return *src; };
                uint32_t s = c * UNIT + sub * 2;
                if (c == 0) {
                    // the unit's first sweep: the run may start inside a line and its first tuples may be carried ones
                    if (s < s_end) {
                        const bool v0 = s >= off, v1 = s + 1 >= off && s + 1 < lim;
                        if (v0 && v1) {
                            const u64 t0 = fetch(s - off), t1 = fetch(s + 1 - off);
                            k6_store16(out64 + base + s, make_uint4((uint32_t)t0, (uint32_t)(t0 >> 32), (uint32_t)t1, (uint32_t)(t1 >> 32)));
                        } else if (v0) k6_store8<NTP>(out64 + base + s, fetch(s - off));
                        else if (v1) k6_store8<NTP>(out64 + base + s + 1, fetch(s + 1 - off));
                    }
                    s += 32;
                }
                // every later slot (s >= 32 > off + carried) holds fresh tuples, neighbours in the sorted tile: two
                // adjacent LDS words, one 16-byte store, no predicates
                const u64 *__restrict__ src = stage + src0 - off - fc;
                for (; s + 1 < s_end; s += 32) {
                    const u64 t0 = src[s], t1 = src[s + 1];
                    k6_store16(out64 + base + s, make_uint4((uint32_t)t0, (uint32_t)(t0 >> 32), (uint32_t)t1, (uint32_t)(t1 >> 32)));
                }
                if (s < s_end) k6_store8<NTP>(out64 + base + s, src[s]);            // the run ends on an even slot
            };
            const uint32_t nunits = F + wsum[NW + 1];
            for (uint32_t u = gid; u < nunits; u += NG) {
                if (u < F) move_unit(u, 0);
                else { const uint32_t h = heavy[u - F]; move_unit(h & 0xFFFFu, h >> 16); }
            }
            if (!RANGED && a.aligned_claims) {
                // the runs' tails (< 16 tuples each): 16 lanes per partition, 8-byte stores
                for (uint32_t idx = tid; idx < F * LINE; idx += BLOCK) {
                    const uint32_t p = idx / LINE, j = idx % LINE;
                    const u64 ti = tinfo[p];
                    if (j < ((uint32_t)ti & (LINE - 1))) k6_store8<NTP>(out64 + (ti >> 4) + j, stage[hist[p] + (meta[p] & 0xFFFFu) + j]);
                }
            }
        } else {
            // ---- stream out: lane i writes tuple i, runs are contiguous ---------------
            for (uint32_t i = tid; i < tile_count; i += BLOCK) {
                const u64 kv = stage[i];
                const uint32_t k = (uint32_t)kv;
                const u64 d = delta[hj_hash(k, factor, F)] + i;
                // (separate output columns - hjgpu_partition, pass 0 of a grouped plan: consecutive lanes write consecutive words, whole
                // lines leave a wave in one instruction; non-temporal like every whole-line store of K6)
#if HJ_K6_STORE == 1
                __builtin_nontemporal_store(k, &a.kout[d]); __builtin_nontemporal_store((uint32_t)(kv >> 32), &a.vout[d]);
#else
                a.kout[d] = k; a.vout[d] = (uint32_t)(kv >> 32);
#endif
            }
        }
        hj_barrier_lds();
        stamp(4);
        if (!nxt.valid) break;
        cur = nxt;
        have_left = true;
        have_left_or_prev = true;
    }
}

// ---- geometry of a pass: workgroup size, vectors per thread, whole-line mode -------
static size_t scatter_lds(int block, int vpt, uint32_t F, bool carry, bool pass2 = false)
{
    const size_t Fpad = (F + 3) & ~3u;
    return Fpad * 16 + (size_t)block * vpt * 4 * 8 + (carry ? Fpad * (HJ_LINE_TUPLES * 8 + 4) : 0) + (pass2 ? Fpad * 8 : 0) +
           (block / 64 + 6) * 4 + HJ_MAX_HEAVY * 4 + 16;
}
constexpr size_t HJ_LDS_LIMIT = 160 * 1024;          // gfx950: 160 KiB per CU, all of it usable by one workgroup

// pass 1: the largest tile whose stage + carry buffers fit the LDS (F <= 209: 16384 tuples,
// <= 421: 12288, <= 640: 8192); beyond that, and for separate output columns, no carry.
// pass 2: 16384-tuple tiles, shared atomic cursors (a workgroup visits a segment about once,
// so there is no "next tile" to complete a line).
// options "scatter_cfg" / "scatter2_cfg" = "block,vpt[,carry]" override (tuning).
ScatterConfig hj_scatter_config(const HjTuning &t, int pass, uint32_t F, bool out_packed)
{
    ScatterConfig cfg = {1024, 4, false};
    const int *o = t.scatter_cfg[pass == 1 ? 0 : 1];
    if (o[0] > 0) {
        cfg.block = o[0]; cfg.vpt = o[1];
        cfg.carry = pass == 1 && out_packed && o[2] != 0 && scatter_lds(o[0], o[1], F, true) <= HJ_LDS_LIMIT;
        return cfg;
    }
    if (pass == 1 && out_packed) {
        for (int vpt = 4; vpt >= 2; --vpt)
            if (scatter_lds(1024, vpt, F, true) <= HJ_LDS_LIMIT) { cfg.vpt = vpt; cfg.carry = true; break; }
    }
    return cfg;
}

int hj_scatter_tile(const HjTuning &t, int pass, uint32_t F, bool out_packed)
{
    const ScatterConfig c = hj_scatter_config(t, pass, F, out_packed);
    return c.block * c.vpt * 4;
}

template <int BLOCK, int VPT, bool RANGED, bool IN_PACKED, bool OUT_PACKED, bool CARRY, bool NTP = true>
static int launch_scatter_t(const ScatterArgs &a, bool want_prof, int cus, hipStream_t stream)
{
    const size_t lds = scatter_lds(BLOCK, VPT, a.F, CARRY, !RANGED);
    if (lds > HJ_LDS_LIMIT) return HJGPU_EINVAL;
    static HjPerDeviceOnce once;
    if (hj_allow_dynamic_lds(reinterpret_cast<const void *>(&scatter_kernel<BLOCK, VPT, RANGED, IN_PACKED, OUT_PACKED, CARRY, NTP>),
                             (int)HJ_LDS_LIMIT, &once) != HJGPU_OK)
        return HJGPU_EHIP;
    // persistent grid: as many workgroups per CU as LDS (160 KiB) and threads (2048) allow
    int per_cu = (int)(HJ_LDS_LIMIT / (lds + 512));
    if (per_cu > 2048 / BLOCK) per_cu = 2048 / BLOCK;
    if (per_cu < 1) per_cu = 1;
    const int grid = cus * per_cu;              // `cus` is what the caller wants occupied (option "reserve_cus")
    ScatterArgs b = a;
    // pass-2 tiles are claimed in order, so all workgroups sit inside the same one or two pass-1
    // partitions, whose output region stays in the Infinity Cache (contiguous ownership instead:
    // 4.45-4.65 vs 3.9-4.0 ms in the first version)
    // diagnostics only: option "scatter_prof" prints where a workgroup's time goes (synchronises!)
    u64 *prof = nullptr;
    b.prof = nullptr;
    if (want_prof) {
        if (hipMalloc(&prof, 8 * sizeof(u64)) != hipSuccess) return HJGPU_ENOMEM;
        (void)hipMemsetAsync(prof, 0, 8 * sizeof(u64), stream);
        b.prof = prof;
    }
    hipLaunchKernelGGL((scatter_kernel<BLOCK, VPT, RANGED, IN_PACKED, OUT_PACKED, CARRY, NTP>), dim3(grid), dim3(BLOCK), lds, stream, b);
    if (b.prof) {
        u64 h[8];
        (void)hipMemcpyAsync(h, prof, sizeof(h), hipMemcpyDeviceToHost, stream);
        (void)hipStreamSynchronize(stream);
        const double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4]);
        fprintf(stderr, "scatter<%d,%d,%s%s> F=%u grid=%d: zero %.1f%% rank(+load wait) %.1f%% scan %.1f%% sort %.1f%% stream-out %.1f%%  (%.0f ticks/wg)\n",
                BLOCK, VPT, RANGED ? "ranged" : "atomic", CARRY ? ",carry" : "", a.F, grid, 100 * h[0] / tot, 100 * h[1] / tot,
                100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, tot / grid);
        (void)hipFree(prof);
    }
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// pass 1: separate columns in, ranged; out packed (join pipeline, whole-line mode when it
//         fits) or separate (hjgpu_partition)
// pass 2: packed in, packed out, atomic cursors
#define SCATTER_CASE(B, V)                                                                   \
    if (c.block == B && c.vpt == V) {                                                        \
        if (a.ranged && !a.in_packed && a.out_packed && c.carry)                                                                                            \
            return a.nt_partial ? launch_scatter_t<B, V, true, false, true, true, true>(a, t.scatter_prof, cus, stream)                                     \
                                : launch_scatter_t<B, V, true, false, true, true, false>(a, t.scatter_prof, cus, stream);                                    \
        if (a.ranged && !a.in_packed && a.out_packed)                                                                                                        \
            return a.nt_partial ? launch_scatter_t<B, V, true, false, true, false, true>(a, t.scatter_prof, cus, stream)                                    \
                                : launch_scatter_t<B, V, true, false, true, false, false>(a, t.scatter_prof, cus, stream);                                   \
        if (a.ranged && !a.in_packed && !a.out_packed) return launch_scatter_t<B, V, true, false, false, false>(a, t.scatter_prof, cus, stream); \
        if (!a.ranged && a.in_packed && a.out_packed)                                                                                                        \
            return a.nt_partial ? launch_scatter_t<B, V, false, true, true, false, true>(a, t.scatter_prof, cus, stream)                                    \
                                : launch_scatter_t<B, V, false, true, true, false, false>(a, t.scatter_prof, cus, stream);                                   \
        return HJGPU_EINVAL;                                                                 \
    }

int hj_launch_scatter(const ScatterArgs &a, const HjTuning &t, int cus, hipStream_t stream)
{
    if (a.F == 0 || a.F > HJGPU_MAX_FANOUT) return HJGPU_EINVAL;
    const ScatterConfig c = hj_scatter_config(t, a.ranged ? 1 : 2, a.F, a.out_packed != 0);
    if (a.ranged && a.geom.tile != (uint32_t)(c.block * c.vpt * 4)) return HJGPU_EINVAL;
    SCATTER_CASE(1024, 4)
    SCATTER_CASE(1024, 3)
    SCATTER_CASE(1024, 2)
    SCATTER_CASE(512, 4)
    SCATTER_CASE(512, 3)
    SCATTER_CASE(512, 2)
    SCATTER_CASE(256, 4)
    SCATTER_CASE(256, 2)
    return HJGPU_EINVAL;
}

// ---- per-context tuning (hj_internal.hpp) --------------------------------------------------------
int hj_allow_dynamic_lds(const void *kernel, int bytes, HjPerDeviceOnce *once)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return HJGPU_EHIP;
    const bool tracked = dev >= 0 && dev < (int)sizeof(once->done);
    if (tracked && __atomic_load_n(&once->done[dev], __ATOMIC_ACQUIRE)) return HJGPU_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return HJGPU_EHIP;
    if (tracked) __atomic_store_n(&once->done[dev], (unsigned char)1, __ATOMIC_RELEASE);
    return HJGPU_OK;
}

static bool parse_flag(const char *v, bool *out)
{
    if (!v) return false;
    char *end = nullptr;
    const long x = strtol(v, &end, 10);
    if (end == v || *end) return false;
    *out = x != 0;
    return true;
}

bool hj_tuning_set(HjTuning *t, const char *name, const char *value)
{
    if (!t || !name || !value) return false;
    auto is = [&](const char *n) { return strcmp(name, n) == 0; };
    if (is("dense2")) return parse_flag(value, &t->dense2);
    if (is("npj_refhash")) return parse_flag(value, &t->npj_refhash);
    if (is("no_broadcast")) return parse_flag(value, &t->no_broadcast);
    if (is("force_chained")) return parse_flag(value, &t->force_chained);
    if (is("scatter_prof")) return parse_flag(value, &t->scatter_prof);
    if (is("unique")) return parse_flag(value, &t->unique);
    if (is("merged_plan")) return parse_flag(value, &t->merged_plan);
    if (is("piece_interleave")) return parse_flag(value, &t->piece_interleave);
    if (is("group_always")) return parse_flag(value, &t->group_always);
    if (is("group_device")) return parse_flag(value, &t->group_device);
    if (is("group_slack")) {
        char *end = nullptr;
        const long x = strtol(value, &end, 10);
        if (end == value || *end || x < 0 || x > 1000) return false;
        t->group_slack = (int)x;
        return true;
    }
    if (is("placement_log")) return parse_flag(value, &t->placement_log);
    if (is("audit")) return parse_flag(value, &t->audit);
    if (is("solo")) return parse_flag(value, &t->solo);
    if (is("hist_min_lds")) {
        char *end = nullptr;
        const long x = strtol(value, &end, 10);
        if (end == value || *end || x < 0 || x > 140 * 1024) return false;
        t->hist_min_lds = (int)x;
        return true;
    }
    if (is("placement")) {
        char *end = nullptr;
        const long x = strtol(value, &end, 10);
        if (end == value || *end || x < 1 || x > 16) return false;
        t->placement = (int)x;
        return true;
    }
    if (is("placement_ms")) {
        char *end = nullptr;
        const long x = strtol(value, &end, 10);
        if (end == value || *end || x < 0 || x > 600000) return false;
        t->placement_ms = (int)x;
        return true;
    }
    if (is("reserve_cus")) {
        char *end = nullptr;
        const long x = strtol(value, &end, 10);
        if (end == value || *end || x < 0 || x > 128) return false;
        t->reserve_cus = (int)x;
        return true;
    }
    if (is("host_batch")) {
        char *end = nullptr;
        const long long x = strtoll(value, &end, 10);
        if (end == value || *end || x < -1) return false;
        t->host_batch = x;
        return true;
    }
    if (is("group_from") || is("group_inner")) {
        char *end = nullptr;
        const long long x = strtoll(value, &end, 10);
        if (end == value || *end || x < 0 || (is("group_inner") && x < 1)) return false;
        (is("group_from") ? t->group_from : t->group_inner) = x;
        return true;
    }
    if (is("batch_tuples")) {
        char *end = nullptr;
        const long long x = strtoll(value, &end, 10);
        if (end == value || *end) return false;
        t->batch_tuples = x;
        return true;
    }
    if (is("range_tiles")) {
        char *end = nullptr;
        const long x = strtol(value, &end, 10);
        if (end == value || *end || x < 0 || x > (1 << 20)) return false;
        t->range_tiles = (int)x;
        return true;
    }
    if (is("join_cfg")) {
        int b, l, u;
        if (!*value) { t->join = JoinConfig{512, 13, 2}; return true; }
        if (sscanf(value, "%d,%d,%d", &b, &l, &u) != 3) return false;
        t->join = JoinConfig{b, l, u};
        return true;
    }
    if (is("scatter_cfg") || is("scatter2_cfg")) {
        int *o = t->scatter_cfg[is("scatter_cfg") ? 0 : 1];
        int b, v, c = -1;
        if (!*value) { o[0] = 0; o[1] = 0; o[2] = -1; return true; }
        if (sscanf(value, "%d,%d,%d", &b, &v, &c) < 2 || b <= 0 || v <= 0) return false;
        o[0] = b; o[1] = v; o[2] = c;
        return true;
    }
    return false;
}

void hj_tuning_from_env(HjTuning *t)
{
    static const char *const names[] = {"dense2", "npj_refhash", "no_broadcast", "force_chained", "scatter_prof",
                                        "unique", "merged_plan", "piece_interleave", "range_tiles", "batch_tuples", "group_from", "group_inner", "group_always", "group_device", "group_slack", "host_batch", "placement", "placement_ms", "placement_log", "audit", "solo", "hist_min_lds", "reserve_cus", "join_cfg", "scatter_cfg", "scatter2_cfg"};
    for (const char *n : names) {
        char env[64] = "HJGPU_";
        size_t at = strlen(env);
        for (const char *c = n; *c && at + 1 < sizeof(env); ++c) env[at++] = (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
        env[at] = 0;
        const char *v = getenv(env);
        if (v && *v && !hj_tuning_set(t, n, v))
            fprintf(stderr, "hjgpu: ignoring malformed %s=%s\n", env, v);      // once per context creation, never on a launch path
    }
}
