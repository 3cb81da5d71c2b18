// join_kernels.hip — K7+K8: per-partition build + probe with the hash table in LDS.
//
// Replaces build()/probe() of phj.cpp:307-397 / 399-571 (scalar definitions
// 577-647) and the join loop phj.cpp:1869-1924 / cpra2.cpp:1883-1971.
// Design (not a translation):
//   * The reference keeps a ~128 KB table per thread in L2 (phj.cpp:1976-1977);
//     here each 512-thread workgroup owns a 64 KiB open-addressing table in LDS
//     (8192 slots of {key, payload}), two workgroups per CU so that one
//     workgroup's table fill/build overlaps the other's probe stream.
//   * Double hashing like the reference, but over a power-of-two table with an
//     odd step: slot = top bits of key*tf0, step = (top bits of key*tf1) | 1.
//     Load factor <= 0.5 by construction (at most SLOTS/2 build tuples per fill;
//     larger partitions are processed in several fills, re-streaming the probe
//     slice — the overflow path for skewed / duplicate-heavy build sides).
//   * Insert = LDS compare-and-swap on the key word against the empty sentinel
//     (the reference serialises lane conflicts by scatter/gather-back,
//     phj.cpp:349-353); the sentinel of partition q is the smallest value that
//     does NOT hash to q (generalises phj.cpp:1886-1897), so key 0 is legal.
//   * Probe streams the S slice with aligned 16-byte loads, four independent
//     chains per lane, walking to the first empty slot and reporting every
//     match (no _UNIQUE, phj.cpp:616-644).
//   * A work item is (partition, slice of its probe rows); CPRA's per-chunk
//     pieces (cpra2.cpp:1891-1959 memcpy gather) are walked in place: the
//     "gather" is just the loop over chunk offsets.
//   * Results: register aggregates (count + 3 sums) reduced per workgroup, or
//     materialised rows through per-wave 64-bit cursors into atomically claimed
//     blocks (the reference's block protocol, npj.cpp:244-246, 312-316).
#include "hj_device.hpp"
#include "hj_internal.hpp"
#include "hj_emit.hpp"

template <int BLOCK, int LOG2SLOTS>
__global__ __launch_bounds__(BLOCK) void join_kernel(JoinArgs a)
{
    constexpr uint32_t SLOTS = 1u << LOG2SLOTS;
    constexpr uint32_t MASK = SLOTS - 1;
    constexpr uint32_t CAP = SLOTS / 2;
    constexpr int SHIFT = 32 - LOG2SLOTS;
    constexpr int NW = BLOCK / 64;
    __shared__ uint2 tab[SLOTS];                 // .x = key, .y = build payload
    __shared__ u64 red[4][NW];
    __shared__ u64 wave_cursor[NW];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const uint32_t P = a.P, C = a.chunks;
    const u64 total_items = a.slice_prefix[P];
    const uint4 *__restrict__ sk4 = reinterpret_cast<const uint4 *>(a.sk - a.s_align);
    const uint4 *__restrict__ sv4 = reinterpret_cast<const uint4 *>(a.sv - a.s_align);
    const uint32_t tf0 = a.tf0, tf1 = a.tf1;

    Emitter em;
    em.init(a.ok, a.oov, a.oiv, a.block_size, a.block_limit, a.block_counter, a.overflow,
            &wave_cursor[wave]);
    if (hj_lane() == 0) wave_cursor[wave] = HJ_NO_CURSOR;

    u64 acc_n = 0, acc_k = 0, acc_o = 0, acc_i = 0;

    for (u64 w = blockIdx.x; w < total_items; w += gridDim.x) {
        const uint32_t q = hj_find_segment(a.slice_prefix, P, w);
        const u64 slice = w - a.slice_prefix[q];
        const u64 nslices = a.slices[q];

        // empty sentinel: smallest value whose partition is not q (P >= 2)
        uint32_t empty = 0;
        while (hj_part2(empty, a.f1, a.F1, a.f2, a.F2) == q) ++empty;

        // total build rows of q over all chunks
        u64 nr = 0;
        for (uint32_t c = 0; c < C; ++c) nr += a.roff[(u64)c * P + q + 1] - a.roff[(u64)c * P + q];

        for (u64 fill_beg = 0; fill_beg < nr; fill_beg += CAP) {
            const u64 fill_end = min(nr, fill_beg + CAP);
            // ---- clear --------------------------------------------------------
            for (uint32_t i = tid; i < SLOTS; i += BLOCK) tab[i] = make_uint2(empty, 0u);
            __syncthreads();
            // ---- build: rows [fill_beg, fill_end) of the chunk-concatenated R_q --
            u64 seen = 0;
            for (uint32_t c = 0; c < C; ++c) {
                const u64 b = a.roff[(u64)c * P + q], e = a.roff[(u64)c * P + q + 1];
                const u64 len = e - b;
                // intersection of [seen, seen+len) with [fill_beg, fill_end)
                const u64 lo = max(seen, fill_beg), hi = min(seen + len, fill_end);
                for (u64 i = lo + tid; i < hi; i += BLOCK) {
                    const u64 row = b + (i - seen);
                    const uint32_t k = a.rk[row];
                    const uint32_t v = a.rv[row];
                    uint32_t slot = (k * tf0) >> SHIFT;
                    const uint32_t step = ((k * tf1) >> SHIFT) | 1u;
                    for (;;) {
                        const uint32_t old = atomicCAS(&tab[slot].x, empty, k);   // ds_cmpst_rtn_b32
                        if (old == empty) { tab[slot].y = v; break; }
                        slot = (slot + step) & MASK;
                    }
                }
                seen += len;
            }
            __syncthreads();
            // ---- probe: this item's share of every chunk piece of S_q -----------
            for (uint32_t c = 0; c < C; ++c) {
                const u64 b = a.soff[(u64)c * P + q], e = a.soff[(u64)c * P + q + 1];
                const u64 len = e - b;
                if (len == 0) continue;
                // sub-range `slice` of `nslices` equal parts (128-bit safe: len < 2^40, slices < 2^24)
                const u64 sb = b + (len * slice) / nslices;
                const u64 se = b + (len * (slice + 1)) / nslices;
                if (se <= sb) continue;
                const u64 gb = a.s_align + sb, ge = a.s_align + se;
                u64 g = (gb & ~3ull) + (u64)tid * 4;
                bool have = g < ge;
                uint4 kk = make_uint4(0, 0, 0, 0), vv = kk;
                if (have) { kk = sk4[g >> 2]; vv = sv4[g >> 2]; }
                while (have) {
                    // prefetch the next vector before walking the chains of this one
                    const u64 g2 = g + (u64)BLOCK * 4;
                    const bool have2 = g2 < ge;
                    uint4 kk2 = make_uint4(0, 0, 0, 0), vv2 = kk2;
                    if (have2) { kk2 = sk4[g2 >> 2]; vv2 = sv4[g2 >> 2]; }

                    const uint32_t key[4] = {kk.x, kk.y, kk.z, kk.w};
                    const uint32_t val[4] = {vv.x, vv.y, vv.z, vv.w};
                    uint32_t slot[4], step[4];
                    uint2 t[4];
                    bool act[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        act[j] = (g + j >= gb) && (g + j < ge);
                        slot[j] = (key[j] * tf0) >> SHIFT;
                        step[j] = ((key[j] * tf1) >> SHIFT) | 1u;
                        t[j] = act[j] ? tab[slot[j]] : make_uint2(empty, 0u);
                    }
                    while (act[0] | act[1] | act[2] | act[3]) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (act[j]) {
                                if (t[j].x == empty) {
                                    act[j] = false;
                                } else {
                                    if (t[j].x == key[j]) {
                                        acc_n += 1; acc_k += key[j]; acc_o += val[j]; acc_i += t[j].y;
                                        em.emit(key[j], val[j], t[j].y);
                                    }
                                    slot[j] = (slot[j] + step[j]) & MASK;
                                    t[j] = tab[slot[j]];
                                }
                            }
                        }
                    }
                    g = g2; have = have2; kk = kk2; vv = vv2;
                }
            }
            __syncthreads();   // table is reused by the next fill / work item
        }
    }

    // ---- per-wave cursors -> final offsets (close_gaps input) ---------------------
    if (a.ok && hj_lane() == 0)
        a.final_offsets[(u64)blockIdx.x * NW + wave] = wave_cursor[wave];

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

int hj_join_grid(int cus) { return cus * 2; }
int hj_join_workers(int cus) { return hj_join_grid(cus) * (HJ_JOIN_BLOCK / 64); }

int hj_launch_join(const JoinArgs &a, int cus, hipStream_t stream)
{
    if (a.P < 2 || a.chunks == 0) return HJGPU_EINVAL;
    hipLaunchKernelGGL((join_kernel<HJ_JOIN_BLOCK, HJ_JOIN_LOG2SLOTS>), dim3(hj_join_grid(cus)),
                       dim3(HJ_JOIN_BLOCK), 0, stream, a);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}
