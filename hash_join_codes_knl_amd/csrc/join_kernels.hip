// join_kernels.hip — K7+K8: per-partition build + probe with the hash table in LDS.
//
// Replaces build()/probe() of phj.cpp:307-397 / 399-571 (scalar definitions
// 577-647) and the join loop phj.cpp:1869-1924 / cpra2.cpp:1883-1971.
// Design (not a translation):
//   * The reference keeps a ~128 KB double-hashing table per thread in L2
//     (phj.cpp:1976-1977); here each workgroup owns an 8192-slot table of
//     {key, payload} words in LDS (64 KiB, two workgroups per CU so that one
//     workgroup's clear/build overlaps the other's probe stream).
//   * FAST PATH - 2-choice cuckoo table.  rocprof PMC showed the open-addressing
//     probe loop was instruction-bound (127 VALU + 77 SALU wave-instructions per
//     probe key: 64 lanes x 4 chains wait for the longest chain), not memory-bound.
//     In a cuckoo table a key lives in exactly one of two slots
//     a1 = top bits of key*tf0, a2 = a1 + odd offset from key*tf1, so a probe is two
//     independent ds_read_b64 and two compares: no loop, no divergence, and up
//     to two copies of a build key are reported naturally (multi-match).
//     Build = ds_wrxchg_rtn_b64 eviction walk, bounded; at load <= 0.5 it succeeds
//     with overwhelming probability for unique keys.
//   * FALLBACK - double-hashing chains (the reference's scheme over a power-of-two
//     table, odd step) when the cuckoo build does not converge: >= 3 copies of a
//     build key (config-1-like duplicate-heavy build sides) or an unlucky cycle.
//     Probe walks to the first empty slot and reports every match (no _UNIQUE,
//     phj.cpp:616-644).
//   * At most SLOTS/2 build tuples per table fill; larger partitions are processed
//     in several fills, re-streaming the probe slice (skew overflow path).
//   * The empty sentinel of partition q is the smallest value that does NOT hash to
//     q (generalises phj.cpp:1886-1897), so key 0 is legal.
//   * A work item is (partition, slice of its probe rows); CPRA's per-chunk pieces
//     (cpra2.cpp:1891-1959 memcpy gather) are walked in place: the "gather" is
//     just the loop over chunk offsets.
//   * Results: register aggregates (count + 3 sums) reduced per workgroup, or
//     materialised rows through per-wave 64-bit cursors into atomically claimed
//     blocks (the reference's block protocol, npj.cpp:244-246, 312-316).
#include "hj_device.hpp"
#include "hj_internal.hpp"
#include "hj_emit.hpp"

// PACKED: the relations arrive as payload << 32 | key tuples (the library's own
// partition passes); !PACKED: separate key / payload columns (hjgpu_join_partitions).
// UNIQUE: the reference's _UNIQUE build (npj.cpp:288-290, phj.cpp:459, 635): a probe tuple reports its FIRST
// match only.  Inside one table that is "one of the key's two cuckoo slots" / "the first hit of the chain";
// a build partition that takes several table fills keeps one bit per probe row of the work item in LDS
// (`matched`), so that a row reported by an earlier fill is skipped by the later ones (such partitions
// are then planned as ONE fill group: all fills of a probe slice stay with one workgroup).
// Round 4: that bookkeeping is needed by multi-fill partitions only - skew, never the planned case - but its code cost
// every _UNIQUE join 26 VGPRs (one probe vector per lane, no build-row prefetch: join phase 1.88 instead of 1.59 ms at
// 64 M x 1 G).  A _UNIQUE join is therefore TWO launches: <UNIQUE, !DEDUP> takes the work items whose build rows fit
// one fill, with the default instance's geometry (two vectors per lane, prefetch, no `matched` words), and skips the
// others; <UNIQUE, DEDUP> takes exactly those and returns at once when the plan counted none (JoinArgs::multi_fill).
// Waves per SIMD the geometry runs at: workgroups per CU (LDS: one table + ~10 KiB each in 160 KiB; threads: 2048
// per CU) x waves per workgroup / 4 SIMDs.  It is the second argument of __launch_bounds__ (HIP: minimum waves per
// execution unit), i.e. the register budget: 512 / waves VGPRs per lane.  More bytes in flight per lane at the price
// of fewer waves does NOT pay here (round 3, profiles/r03_ab_emit.txt: 384 threads x 4 vectors at 3 waves per SIMD
// 3.15 ms, 256 x 8 at 2 waves 2.74 ms, against 1.58 ms for 512 x 2 at 4 waves): K7+K8 is bound by instruction issue
// and LDS latency, which only resident waves hide, not by the probe stream's bytes in flight.
constexpr int hj_join_wgs_per_cu(int block, int log2slots)
{
    const int by_lds = (160 * 1024) / ((1 << log2slots) * 8 + 1024 + 9 * 1024), by_threads = 2048 / block;
    const int n = by_lds < by_threads ? by_lds : by_threads;
    return n < 1 ? 1 : n;
}
constexpr int hj_join_waves_per_simd(int block, int log2slots)
{
    const int w = hj_join_wgs_per_cu(block, log2slots) * (block / 64) / 4;
    return w < 1 ? 1 : w;
}

// HJ_EMIT4 (build-time, default 1; 0 for A/B): a probe vector whose four tuples matched exactly once each leaves the lane as ONE 16-byte
// store per result column (EmitterT::emit4) instead of four 4-byte ones.  Round 6, 64 M x 1 G with 10^9 rows, default policy
// (non-temporal rows), one process, same allocations: join 4.48 -> 3.83 ms (profiles/r06_ab_emit4.txt) - the 4-byte non-temporal
// stores cost 14 % over plain ones (round 5), whole 16-byte pieces cost nothing, as in K6.
#ifndef HJ_EMIT4
#define HJ_EMIT4 1
#endif
// NTROWS: result rows through non-temporal stores (EmitterT<true>; JoinArgs::nt_rows) - false only for solo joins
template <int BLOCK, int LOG2SLOTS, int BATCH, bool PACKED, bool UNIQUE, bool DEDUP = false, bool NTROWS = true>
__global__ __launch_bounds__(BLOCK, hj_join_waves_per_simd(BLOCK, LOG2SLOTS)) void join_kernel(JoinArgs a)
{
    static_assert(UNIQUE || !DEDUP, "DEDUP is the multi-fill half of a _UNIQUE join");
    // the plan found no partition that takes several fills (the planned case): nothing for this launch to do
    if (DEDUP && a.multi_fill && *a.multi_fill == 0) return;
    constexpr uint32_t SLOTS = 1u << LOG2SLOTS;
    constexpr uint32_t MASK = SLOTS - 1;
    constexpr uint32_t CAP = SLOTS / 2;
    constexpr int SHIFT = 32 - LOG2SLOTS;
    constexpr int NW = BLOCK / 64;
    constexpr int RB = 8;                        // build rows a lane loads before inserting
    constexpr int CUCKOO_MAX_EVICTIONS = 64;
    __shared__ u64 tab64[SLOTS];                 // low word = key, high word = build payload
    __shared__ u64 red[4][NW];
    __shared__ u64 wave_cursor[NW];
    __shared__ uint32_t cuckoo_failed;
    // UNIQUE: one bit per probe row of the current work item (<= HJ_JOIN_SLICE + 1 rows per chunk piece)
    constexpr uint32_t MATCHED_WORDS = DEDUP ? (HJ_JOIN_SLICE + 64) / 32 + 2 : 1;
    __shared__ uint32_t matched[MATCHED_WORDS];
    constexpr bool dedup = DEDUP;                // every item of the DEDUP launch takes more than one fill, none of the other's
    uint2 *tab = reinterpret_cast<uint2 *>(tab64);   // chained view: .x = key, .y = payload

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const uint32_t P = a.P, C = a.chunks;
    const u64 total_items = a.slice_prefix[P];
    const uint4 *__restrict__ sk4 = reinterpret_cast<const uint4 *>(PACKED ? a.sk : a.sk - a.s_align);
    const uint4 *__restrict__ sv4 = reinterpret_cast<const uint4 *>(PACKED ? a.sk : a.sv - a.s_align);
    const u64 *__restrict__ r64 = reinterpret_cast<const u64 *>(a.rk);
    const uint32_t tf0 = a.tf0, tf1 = a.tf1;

    EmitterT<NTROWS> em;
    em.init(a.ok, a.oov, a.oiv, a.block_size, a.block_limit, a.block_counter, a.overflow,
            &wave_cursor[wave]);
    // (the multi-fill half of a _UNIQUE join runs behind the single-fill half on the same stream, with the same grid: wave w of
    // workgroup b goes on in the block that wave w of workgroup b left open, and leaves its cursor in the same slot - one
    // launch's worth of worker slots for close_gaps, not two)
    // (device-planned groups - JoinArgs::resume - do the same from group to group: one block counter, one set of open blocks, one close_gaps)
    if (hj_lane() == 0) wave_cursor[wave] = ((DEDUP || a.resume) && a.ok) ? a.final_offsets[(u64)blockIdx.x * NW + wave] : HJ_NO_CURSOR;

    u64 acc_n = 0, acc_k = 0, acc_o = 0, acc_i = 0;
    uint32_t empty = 0;
    uint32_t q = 0;

    // ---- visit rows [fill_beg, fill_end) of the chunk-concatenated build partition q ----
    // rows below `from_row` were already inserted (from the prefetch registers)
    auto for_each_build_row = [&](u64 fill_beg, u64 fill_end, u64 from_row, auto insert) {
        if (C > 1) {
            // chunked relation (CPRA): the partition's rows lie in C pieces.  Walking the pieces one after
            // another costs one memory round trip per piece (C x ~2 us against ~40 us per work item); here a
            // lane's RB rows are rows of the CONCATENATION, so that all loads of a fill are in flight together.
            u64 pb[8], cum[9];                        // piece c = rows [cum[c], cum[c+1]) at pb[c]
            cum[0] = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                pb[c] = 0; cum[c + 1] = cum[c];
                if ((uint32_t)c < C) { pb[c] = a.roff[(u64)c * P + q]; cum[c + 1] = cum[c] + (a.rend[(u64)c * P + q] - pb[c]); }
            }
            for (u64 base = max(fill_beg, from_row); base < fill_end; base += (u64)BLOCK * RB) {
                uint32_t k[RB], v[RB];
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const u64 i = base + (u64)j * BLOCK + tid;
                    k[j] = 0; v[j] = 0;
                    if (i < fill_end) {
                        u64 at = pb[0] + i;
#pragma unroll
                        for (int c = 1; c < 8; ++c) if (i >= cum[c]) at = pb[c] + (i - cum[c]);   // cum is non-decreasing
                        if (PACKED) { const u64 t = r64[at]; k[j] = (uint32_t)t; v[j] = (uint32_t)(t >> 32); }
                        else { k[j] = a.rk[at]; v[j] = a.rv[at]; }
                    }
                }
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const u64 i = base + (u64)j * BLOCK + tid;
                    if (i < fill_end) insert(k[j], v[j]);
                }
            }
            return;
        }
        u64 seen = 0;
        for (uint32_t c = 0; c < C; ++c) {
            const u64 b = a.roff[(u64)c * P + q], e = a.rend[(u64)c * P + q];
            const u64 len = e - b;
            const u64 lo = max(max(seen, fill_beg), from_row), hi = min(seen + len, fill_end);
            for (u64 base = lo; base < hi; base += (u64)BLOCK * RB) {
                uint32_t k[RB], v[RB];
                // all loads of the batch are issued before the first insert
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const u64 i = base + (u64)j * BLOCK + tid;
                    k[j] = 0; v[j] = 0;
                    if (i < hi) {
                        if (PACKED) { const u64 t = r64[b + (i - seen)]; k[j] = (uint32_t)t; v[j] = (uint32_t)(t >> 32); }
                        else { k[j] = a.rk[b + (i - seen)]; v[j] = a.rv[b + (i - seen)]; }
                    }
                }
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const u64 i = base + (u64)j * BLOCK + tid;
                    if (i < hi) insert(k[j], v[j]);
                }
            }
            seen += len;
        }
    };

    // ---- stream the S rows [gb, ge): BATCH key + BATCH payload vectors in flight per lane ----
    // `row0`: index of row gb among the probe rows of this work item (UNIQUE's `matched` bits)
    auto for_each_probe_vector = [&](u64 gb, u64 ge, u64 row0, auto probe4) {
        for (u64 g0 = (gb & ~3ull) + (u64)tid * 4; g0 < ge; g0 += (u64)BLOCK * 4 * BATCH) {
            uint4 kk[BATCH], vv[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const u64 g = g0 + (u64)u * BLOCK * 4;
                kk[u] = make_uint4(0, 0, 0, 0); vv[u] = kk[u];
                if (g < ge) {
                    if (PACKED) { kk[u] = sk4[g >> 1]; vv[u] = sk4[(g >> 1) + 1]; }     // 4 tuples = 2 x 16 bytes
                    else { kk[u] = sk4[g >> 2]; vv[u] = sv4[g >> 2]; }
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const u64 g = g0 + (u64)u * BLOCK * 4;
                if (g >= ge) break;
                const uint32_t key[4] = {kk[u].x, PACKED ? kk[u].z : kk[u].y, PACKED ? vv[u].x : kk[u].z, PACKED ? vv[u].z : kk[u].w};
                const uint32_t val[4] = {PACKED ? kk[u].y : vv[u].x, PACKED ? kk[u].w : vv[u].y, PACKED ? vv[u].y : vv[u].z, vv[u].w};
                bool valid[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    valid[j] = (g + j >= gb) && (g + j < ge);
                    // A probe key that equals the empty sentinel must not "match" empty slots.  In partitioned
                    // joins no tuple of partition q carries that value (it hashes elsewhere); in a broadcast
                    // join the probe side is the caller's unpartitioned column and may hold it (no build key
                    // does).  Only the separate-column instance serves broadcast joins.
                    if (!PACKED) valid[j] = valid[j] && key[j] != empty;
                }
                if (UNIQUE && dedup) {
                    // rows an earlier fill of this partition has already reported are done
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t row = (uint32_t)(row0 + (g + j - gb));
                        if (valid[j]) valid[j] = !((matched[row >> 5] >> (row & 31)) & 1u);
                    }
                    const uint32_t hits = probe4(key, val, valid);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t row = (uint32_t)(row0 + (g + j - gb));
                        if ((hits >> j) & 1u) atomicOr(&matched[row >> 5], 1u << (row & 31));
                    }
                } else (void)probe4(key, val, valid);
            }
        }
    };

    // cuckoo probe: two independent slot reads per key, no loop
    auto probe4_cuckoo = [&](const uint32_t (&key)[4], const uint32_t (&val)[4], const bool (&valid)[4]) -> uint32_t {
        u64 t1[4], t2[4];
        uint32_t hits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a1 = (key[j] * tf0) >> SHIFT;
#if defined(HJ_JOIN_LIMIT_STUDY) && HJ_JOIN_LIMIT_STUDY == 1
            // LIMIT STUDY, never the product (tools/build_variant.py join_one_slot -DHJ_JOIN_LIMIT_STUDY=1; results are WRONG): the second
            // slot's multiply, address and LDS read do not exist - an upper bound on what ANY scheme that fetches a key's two slots with one
            // LDS instruction and one address computation could save (round 5's review asked for ds_read2_b64; its two offsets are
            // immediates of the instruction, the same for every lane, and a fixed distance between a key's slots is no cuckoo table)
            t1[j] = tab64[a1];
            t2[j] = t1[j] ^ ((u64)1 << 63);
#else
            const uint32_t a2 = (a1 + (((key[j] * tf1) >> SHIFT) | 1u)) & MASK;
            t1[j] = tab64[a1];
            t2[j] = tab64[a2];
#endif
        }
        u64 sk_ = 0, so_ = 0, si_ = 0;
        uint32_t n = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool h1 = valid[j] && ((uint32_t)t1[j] == key[j]);
            const bool h2 = valid[j] && ((uint32_t)t2[j] == key[j]) && !(UNIQUE && h1);
            const uint32_t m = (h1 ? 1u : 0u) + (h2 ? 1u : 0u);
            hits |= m ? 1u << j : 0u;
            n += m;
            sk_ += (u64)key[j] * m;
            so_ += (u64)val[j] * m;
            si_ += (h1 ? (uint32_t)(t1[j] >> 32) : 0u);
            si_ += (h2 ? (uint32_t)(t2[j] >> 32) : 0u);
#if !HJ_EMIT4
            if (a.ok) {
                // one emit for "this key matched" (with unique build keys that is every lane of the wave:
                // 64 rows, the cursor moves in whole lines), a second one only for a key found in BOTH slots
                if (h1 | h2) em.emit(key[j], val[j], (uint32_t)((h1 ? t1[j] : t2[j]) >> 32));
                if (h1 & h2) em.emit(key[j], val[j], (uint32_t)(t2[j] >> 32));
            }
#endif
        }
#if HJ_EMIT4
        if (a.ok) {
            // a vector whose four probe tuples matched exactly once each (the common case) leaves as ONE 16-byte store per column and
            // lane (EmitterT::emit4); any other vector row by row as before
            bool once[4], twice = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool h1 = valid[j] && ((uint32_t)t1[j] == key[j]);
                const bool h2 = valid[j] && ((uint32_t)t2[j] == key[j]) && !(UNIQUE && h1);
                once[j] = h1 != h2; twice = twice || (h1 && h2);
            }
            // (a wave's emit4 writes up to 256 rows and claims at most ONE new block: blocks of 512 rows and more)
            if (once[0] && once[1] && once[2] && once[3] && a.block_size >= 512) {
                uint32_t iv4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) iv4[j] = (uint32_t)((((uint32_t)t1[j] == key[j]) ? t1[j] : t2[j]) >> 32);
                em.emit4(key, val, iv4);
            } else if (once[0] || once[1] || once[2] || once[3] || twice) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool h1 = valid[j] && ((uint32_t)t1[j] == key[j]);
                    const bool h2 = valid[j] && ((uint32_t)t2[j] == key[j]) && !(UNIQUE && h1);
                    if (h1 | h2) em.emit(key[j], val[j], (uint32_t)((h1 ? t1[j] : t2[j]) >> 32));
                    if (h1 & h2) em.emit(key[j], val[j], (uint32_t)(t2[j] >> 32));
                }
            }
        }
#endif
        acc_n += n; acc_k += sk_; acc_o += so_; acc_i += si_;
        return hits;
    };

    // chained probe: 4 chains per lane advanced in lock step to the first empty slot
    auto probe4_chained = [&](const uint32_t (&key)[4], const uint32_t (&val)[4], const bool (&valid)[4]) -> uint32_t {
        uint32_t slot[4], step[4];
        uint2 t[4];
        bool live[4];
        uint32_t hits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            live[j] = valid[j];
            slot[j] = (key[j] * tf0) >> SHIFT;
            step[j] = ((key[j] * tf1) >> SHIFT) | 1u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { t[j] = make_uint2(empty, 0u); if (live[j]) t[j] = tab[slot[j]]; }
        for (;;) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool hit = live[j] && (t[j].x == key[j]);
                acc_n += hit ? 1u : 0u;
                acc_k += hit ? key[j] : 0u;
                acc_o += hit ? val[j] : 0u;
                acc_i += hit ? t[j].y : 0u;
                if (a.ok) { if (hit) em.emit(key[j], val[j], t[j].y); }
                hits |= hit ? 1u << j : 0u;
                live[j] = live[j] && (t[j].x != empty) && !(UNIQUE && hit);
                slot[j] = (slot[j] + step[j]) & MASK;
            }
            if (!(live[0] | live[1] | live[2] | live[3])) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (live[j]) t[j] = tab[slot[j]];
        }
        return hits;
    };

    auto probe_item = [&](u64 slice, u64 nslices, auto probe4) {
        u64 row0 = 0;
        for (uint32_t c = 0; c < C; ++c) {
            const u64 b = a.soff[(u64)c * P + q], e = a.send[(u64)c * P + q];
            const u64 len = e - b;
            if (len == 0) continue;
            // sub-range `slice` of `nslices` equal parts (len < 2^40, slices < 2^24)
            const u64 sb = b + (len * slice) / nslices;
            const u64 se = b + (len * (slice + 1)) / nslices;
            if (se <= sb) continue;
            for_each_probe_vector(a.s_align + sb, a.s_align + se, row0, probe4);
            row0 += se - sb;
        }
    };

    // work items are claimed dynamically (one atomic per item): partitions differ in size
    // and so does the memory system's service, a static round-robin leaves a tail
    // Work-item descriptors are double-buffered: while item n runs out of slot `par`, thread 0
    // claims item n+1 into the other slot at the top of the loop; the clear barrier publishes it,
    // and (single-chunk joins) every lane then loads its share of item n+1's build rows into
    // registers right after item n's build, so that they arrive during the probe of item n.
    __shared__ u64 d_item[2], d_rb[2], d_rn[2];
    __shared__ u64 d_rows_beg[2], d_rows_end[2];      // build rows (of the chunk-concatenated partition) this item inserts
    __shared__ uint32_t d_q[2], d_slice[2], d_nslices[2];
    auto claim = [&](int slot) {                          // thread 0 only
        const u64 w = atomicAdd(a.work_counter, 1ull);
        d_item[slot] = w;
        if (w < total_items) {
            const uint32_t nq = a.item_part[w];
            d_q[slot] = nq;
            // item t of partition q = (probe slice t % nslices, fill group t / nslices), see plan_items_kernel;
            // the divisions are done here, by one thread and one item ahead of use
            const u64 shape = a.slices[nq];
            const uint32_t nslices = (uint32_t)shape, groups = (uint32_t)(shape >> 32);
            const uint32_t t = (uint32_t)(w - a.slice_prefix[nq]);
            const u64 rb = a.roff[nq], rn = a.rend[nq] - rb;
            d_rb[slot] = rb;
            d_rn[slot] = rn;
            d_nslices[slot] = nslices;
            u64 rows = rn;
            for (uint32_t c = 1; c < C; ++c) rows += a.rend[(u64)c * P + nq] - a.roff[(u64)c * P + nq];
            if (groups == 1) {
                // the planned case: the partition fits one table (or its few fills stay with one workgroup)
                d_slice[slot] = t;
                d_rows_beg[slot] = 0;
                d_rows_end[slot] = rows;
            } else {
                // this item's share of the oversize partition's table fills
                const uint32_t group = t / nslices;
                d_slice[slot] = t - group * nslices;
                const uint32_t fills = (uint32_t)((rows + CAP - 1) >> (LOG2SLOTS - 1));
                const uint32_t per_group = (fills + groups - 1) / groups;
                d_rows_beg[slot] = min(rows, (u64)group * per_group * CAP);
                d_rows_end[slot] = min(rows, (u64)(group + 1) * per_group * CAP);
            }
        }
    };
    uint32_t pk[RB], pv[RB];                              // prefetched build rows j*BLOCK + tid
#pragma unroll
    for (int j = 0; j < RB; ++j) { pk[j] = 0; pv[j] = 0; }
    u64 pre_rows = 0;                                     // rows [0, pre_rows) of the coming item are in pk/pv
    int par = 0;
    if (tid == 0) claim(0);
    __syncthreads();
    for (;;) {
        const u64 w = hj_uniform(d_item[par]);
        if (w >= total_items) break;
        q = hj_uniform(d_q[par]);
        const u64 slice = hj_uniform(d_slice[par]);
        const u64 nslices = hj_uniform(d_nslices[par]);
        const u64 rows_beg = hj_uniform(d_rows_beg[par]), rows_end = hj_uniform(d_rows_end[par]);
        if (tid == 0) claim(par ^ 1);                     // published by the clear barrier below
        u64 have_rows = pre_rows;
        pre_rows = 0;

        // empty sentinel: smallest value whose partition is not q (P >= 2); broadcast join: a value that no
        // build key equals, found once by broadcast_meta_kernel
        if (a.broadcast) empty = hj_uniform(*a.sentinel);
        else {
            // (pre-partitioned relations: p1_base shifts the pass-1 partition; values of other ranks' partitions wrap to
            // numbers far above P and never equal q)
            empty = 0;
            while ((hj_hash(empty, a.f1, a.F1) - a.p1_base) * a.F2 + hj_hash(empty, a.f2, a.F2) == q) ++empty;
        }
        const u64 EMPTY64 = (u64)empty;
        if (UNIQUE) {
            // single-fill items belong to the <UNIQUE, !DEDUP> launch, multi-fill items to <UNIQUE, DEDUP>
            const bool multi = rows_end - rows_beg > CAP;
            if (multi != DEDUP) {
                __syncthreads();                          // everybody has read this item's slot; publishes the next claim
                par ^= 1;
                continue;
            }
            if (DEDUP) for (uint32_t i = tid; i < MATCHED_WORDS; i += BLOCK) matched[i] = 0;    // published by the clear barrier
        }

        for (u64 fill_beg = rows_beg; fill_beg < rows_end; fill_beg += CAP) {
            const u64 fill_end = min(rows_end, fill_beg + CAP);
            // ---- clear + cuckoo build --------------------------------------------
            for (uint32_t i = tid; i < SLOTS; i += BLOCK) tab64[i] = EMPTY64;
            if (tid == 0) cuckoo_failed = a.force_chained;
            __syncthreads();
            auto cuckoo_insert = [&](uint32_t k, uint32_t v) {
                u64 cur = (u64)k | ((u64)v << 32);
                uint32_t loc = (k * tf0) >> SHIFT;
                int it = 0;
                for (; it < CUCKOO_MAX_EVICTIONS; ++it) {
                    const u64 old = atomicExch(&tab64[loc], cur);            // ds_wrxchg_rtn_b64
                    if ((uint32_t)old == empty) break;                       // slot was free
                    // `old` was evicted: it moves to the other one of its two slots
                    const uint32_t ok_ = (uint32_t)old;
                    const uint32_t a1 = (ok_ * tf0) >> SHIFT;
                    const uint32_t a2 = (a1 + (((ok_ * tf1) >> SHIFT) | 1u)) & MASK;
                    loc = (loc == a1) ? a2 : a1;
                    cur = old;
                }
                if (it == CUCKOO_MAX_EVICTIONS) cuckoo_failed = 1;           // a tuple is left in hand
            };
            const u64 from_regs = (fill_beg == 0) ? have_rows : 0;
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if ((u64)j * BLOCK + tid < from_regs) cuckoo_insert(pk[j], pv[j]);
            for_each_build_row(fill_beg, fill_end, from_regs, cuckoo_insert);
            __syncthreads();
            // build rows of the NEXT item: issue the loads now, they land during this probe
            // (not in the DEDUP instance: the 16 prefetch registers are what it would spill to scratch, and no shipped
            // kernel may use scratch - see the note at hj_launch_join)
            if (!DEDUP && fill_beg == 0 && C == 1 && PACKED && d_item[par ^ 1] < total_items) {
                const u64 nb = d_rb[par ^ 1];
                pre_rows = min(min(d_rn[par ^ 1], (u64)BLOCK * RB), (u64)CAP);
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const u64 i = (u64)j * BLOCK + tid;
                    if (i < pre_rows) { const u64 t = r64[nb + i]; pk[j] = (uint32_t)t; pv[j] = (uint32_t)(t >> 32); }
                }
            }
            if (!cuckoo_failed) {
                probe_item(slice, nslices, probe4_cuckoo);
            } else {
                // ---- fallback: rebuild as double-hashing chains, multi-match probe -----
                __syncthreads();
                for (uint32_t i = tid; i < SLOTS; i += BLOCK) tab64[i] = EMPTY64;
                __syncthreads();
                for_each_build_row(fill_beg, fill_end, 0, [&](uint32_t k, uint32_t v) {
                    uint32_t slot = (k * tf0) >> SHIFT;
                    const uint32_t step = ((k * tf1) >> SHIFT) | 1u;
                    for (;;) {
                        const uint32_t old = atomicCAS(&tab[slot].x, empty, k);   // ds_cmpst_rtn_b32
                        if (old == empty) { tab[slot].y = v; break; }
                        slot = (slot + step) & MASK;
                    }
                });
                __syncthreads();
                probe_item(slice, nslices, probe4_chained);
            }
            __syncthreads();   // table is reused by the next fill / work item
        }
        if (rows_beg >= rows_end) __syncthreads();   // no fill (a trailing fill group of an oversize partition): publish the next claim
        par ^= 1;
    }

    // ---- per-wave cursors -> final offsets (close_gaps input) ---------------------
    if (a.ok && hj_lane() == 0)
        hj_store(&a.final_offsets[(u64)blockIdx.x * NW + wave], wave_cursor[wave]);

    // ---- workgroup reduction of the aggregates, 4 atomics per workgroup ---------
    acc_n = wave_reduce_sum(acc_n); acc_k = wave_reduce_sum(acc_k);
    acc_o = wave_reduce_sum(acc_o); acc_i = wave_reduce_sum(acc_i);
    if (hj_lane() == 0) { red[0][wave] = acc_n; red[1][wave] = acc_k; red[2][wave] = acc_o; red[3][wave] = acc_i; }
    __syncthreads();
    if (tid < 4) {
        u64 s = 0;
        for (int i = 0; i < NW; ++i) s += red[tid][i];
        u64 *dst = reinterpret_cast<u64 *>(a.result) + tid;
        if (s) atomicAdd(dst, s);
    }
}

#include <stdlib.h>
#include <stdio.h>

// Broadcast join metadata (see BroadcastMeta).  One workgroup: a bitmap of the build keys' low 14 bits in LDS
// (inner < 16384 keys cannot occupy all 16384 residues), the first free residue is the sentinel.
__global__ __launch_bounds__(1024) void broadcast_meta_kernel(const uint32_t *__restrict__ keys, u64 inner, u64 outer,
                                                               uint32_t nslices, uint32_t groups, BroadcastMeta m)
{
    __shared__ uint32_t bits[512];
    __shared__ uint32_t first_free;
    for (uint32_t i = threadIdx.x; i < 512; i += 1024) bits[i] = 0;
    if (threadIdx.x == 0) first_free = 0xFFFFFFFFu;
    __syncthreads();
    for (u64 i = threadIdx.x; i < inner; i += 1024) {
        const uint32_t r = keys[i] & 16383u;
        atomicOr(&bits[r >> 5], 1u << (r & 31));
    }
    __syncthreads();
    if (threadIdx.x < 512) {
        const uint32_t free_bits = ~bits[threadIdx.x];
        if (free_bits) atomicMin(&first_free, threadIdx.x * 32 + (uint32_t)__builtin_ctz(free_bits));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        hj_store(m.sentinel, first_free);
        hj_store(&m.roff[0], (u64)0); hj_store(&m.rend[0], inner); hj_store(&m.soff[0], (u64)0); hj_store(&m.send[0], outer);
        hj_store(&m.slice_prefix[0], (u64)0); hj_store(&m.slice_prefix[1], (u64)nslices * groups);
        hj_store(&m.slices[0], (u64)nslices | ((u64)groups << 32));
    }
}

int hj_launch_broadcast_meta(const uint32_t *inner_keys, size_t inner, size_t outer, uint32_t nslices,
                             uint32_t groups, const BroadcastMeta &m, hipStream_t stream)
{
    if (inner == 0 || inner > 16383 || nslices == 0 || groups == 0) return HJGPU_EINVAL;
    hipLaunchKernelGGL(broadcast_meta_kernel, dim3(1), dim3(1024), 0, stream, inner_keys, (u64)inner, (u64)outer,
                       nslices, groups, m);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

const JoinConfig &hj_join_config_big()
{
    static const JoinConfig cfg = {1024, 14, 2};
    return cfg;
}

// workgroups per CU the LDS table allows (160 KiB per CU), capped by 2048 threads per CU
static int join_wgs_per_cu(const JoinConfig &c)
{
    // + 9 KiB: the UNIQUE instances' `matched` bits; the same grid for both keeps hj_join_workers one number
    return hj_join_wgs_per_cu(c.block, c.log2slots);
}

static int join_grid(int cus, const JoinConfig &c) { return cus * join_wgs_per_cu(c); }
// Worker slots (final_offsets entries, one open output block each) of a join: one per wave of ONE launch.  A _UNIQUE join is
// two launches (see join_kernel) of the same grid; the second half's waves continue in the first half's open blocks.
int hj_join_workers(const HjTuning &t, int cus, bool big_tables, bool unique)
{
    (void)unique;
    const JoinConfig &c = hj_join_config_of(t, big_tables);
    return join_grid(cus, c) * (c.block / 64);
}

// (the plain-row instances serve solo materialising joins only; aggregate-only joins never emit: the NTROWS = true instance)
#define JOIN_LAUNCH(B, L, U, P, UNQ, DD, ARGS)                                                                          \
    do {                                                                                                                \
        if ((ARGS).ok && !(ARGS).nt_rows) hipLaunchKernelGGL((join_kernel<B, L, U, P, UNQ, DD, false>), dim3(join_grid(cus, c)), dim3(B), 0, stream, ARGS); \
        else hipLaunchKernelGGL((join_kernel<B, L, U, P, UNQ, DD, true>), dim3(join_grid(cus, c)), dim3(B), 0, stream, ARGS);                              \
    } while (0)
#define JOIN_CASE(B, L, U, UNQ)                                                                   \
    if (c.block == B && c.log2slots == L && c.batch == U && (b.unique != 0) == UNQ) {             \
        if (b.packed) JOIN_LAUNCH(B, L, U, true, UNQ, false, b);                                  \
        else JOIN_LAUNCH(B, L, U, false, UNQ, false, b);                                          \
        return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;                           \
    }

// The geometries that are built (option "join_cfg"): {block, log2slots, batch, a UNIQUE instance exists}.
// NO SHIPPED INSTANCE MAY USE SCRATCH (tests/test_kernel_resources.py reads the compiler's remarks): a K6 instance with a
// private segment loses stores next to other streams' kernels (DESIGN section 3 "Round 4": found by round 3's multi-GPU
// stress runs, narrowed down in round 4 - the private values themselves are never wrong; the cause is below the ISA), and
// nothing says the other kernels would be exempt.  The multi-fill half of a _UNIQUE join (<UNIQUE, DEDUP>) therefore runs
// with ONE probe vector per lane and without the build-row prefetch (what fits 128 VGPRs), and the geometries that
// spill (512,13,1 and 512,13,4 without _UNIQUE) are not built.
static const struct { int block, log2slots, batch; bool unique; } JOIN_BUILT[] = {
    {512, 13, 2, true}, {1024, 14, 2, true}, {256, 12, 2, false},
};

bool hj_join_config_built(const JoinConfig &c, bool unique)
{
    for (const auto &g : JOIN_BUILT)
        if (g.block == c.block && g.log2slots == c.log2slots && g.batch == c.batch && (!unique || g.unique)) return true;
    return false;
}

// the two launches of a _UNIQUE join (see join_kernel): single-fill items at the default geometry, then the multi-fill
// items (their own work counter and worker slots) at one vector per lane
#define JOIN_CASE_UNIQUE(B, L)                                                                    \
    if (c.block == B && c.log2slots == L && b.unique) {                                           \
        JoinArgs d = b;                                                                           \
        d.work_counter = b.work_counter2;                                                         \
        if (b.packed) { JOIN_LAUNCH(B, L, 2, true, true, false, b); JOIN_LAUNCH(B, L, 1, true, true, true, d); }      \
        else { JOIN_LAUNCH(B, L, 2, false, true, false, b); JOIN_LAUNCH(B, L, 1, false, true, true, d); }             \
        return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;                           \
    }

int hj_launch_join(const JoinArgs &a, const HjTuning &t, int cus, hipStream_t stream)
{
    if ((a.P < 2 && !a.broadcast) || a.P < 1 || a.chunks == 0) return HJGPU_EINVAL;
    const JoinConfig &c = hj_join_config_of(t, a.big_tables != 0);
    JoinArgs b = a;
    b.force_chained = t.force_chained ? 1u : 0u;       // tests: exercise the fallback table everywhere
    b.unique = (a.unique || t.unique) ? 1u : 0u;
    if (b.unique && !b.work_counter2) return HJGPU_EINVAL;
    JOIN_CASE_UNIQUE(512, 13)
    JOIN_CASE_UNIQUE(1024, 14)
    JOIN_CASE(512, 13, 2, false)
    JOIN_CASE(1024, 14, 2, false)
    JOIN_CASE(256, 12, 2, false)
    return HJGPU_EINVAL;
}
