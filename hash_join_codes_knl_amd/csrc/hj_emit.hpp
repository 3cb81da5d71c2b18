// hj_emit.hpp — materialised join output: the reference's block protocol
// (npj.cpp:244-246, 312-316, 426-436) at wave granularity.
//
// Every wave owns a 64-bit cursor `o` (kept in LDS so that it stays coherent
// across the wave's divergent control flow).  A block of `block_size` output
// slots is claimed with ONE global atomic (fetch_add on the block counter);
// the lanes that found a match in the same instruction get consecutive slots
// (ballot + mbcnt prefix), so a wave writes its matches as one contiguous run.
// When a run fills the current block the wave claims the next one eagerly,
// exactly like the reference's `if ((o & (block_size-1)) == 0) o = claim()`.
// The first block is claimed lazily (HJ_NO_CURSOR) so idle waves leave no hole.
// The partially filled last block of every wave is compacted by K9
// (close_gaps, npj.cpp:475-514).
#pragma once
#include "hj_device.hpp"

#define HJ_NO_CURSOR (~0ull)
#ifndef HJ_ROW_STORE
#define HJ_ROW_STORE 1
#endif

// The cursor is addressed as LDS (address space 3), not through a generic pointer: with a generic `volatile u64 *`
// the compiler emitted flat_load / flat_store with `s_waitcnt vmcnt(0)` around every emit - FLAT operations count on
// the vector-memory counter, so each emit first waited for EVERY outstanding global load and store of the wave
// (found in round 3; measured effect on the materialising join: none - its rate is set by the row stores themselves,
// profiles/r03_ab_emit.txt - but an emit no longer serialises the wave's memory pipeline).
typedef __attribute__((address_space(3))) u64 hj_lds_u64;

// NT: the rows leave through non-temporal stores (EmitterT<true>, every join that may run beside other work) or plain ones
// (EmitterT<false>: solo joins, option "solo").  A template parameter: a run-time flag around the three stores was merged by the
// compiler into ONE plain store per column (tests/test_store_policy_isa.py reads the machine code).
template <bool NT>
struct EmitterT {
    uint32_t *ok, *oov, *oiv;
    u64 block_size, block_limit;
    u64 *block_counter;
    uint32_t *overflow;
    volatile hj_lds_u64 *cursor;   // this wave's cursor, in LDS (ds_read_b64 / ds_write_b64)

    __device__ __forceinline__ void init(uint32_t *k, uint32_t *ov, uint32_t *iv, u64 bs, u64 bl,
                                         u64 *bc, uint32_t *ovf, u64 *lds_cursor)
    {
        ok = k; oov = ov; oiv = iv; block_size = bs; block_limit = bl;
        block_counter = bc; overflow = ovf; cursor = (volatile hj_lds_u64 *)lds_cursor;
    }

    // Called by the lanes that have a match (any subset of the wave).
    __device__ __forceinline__ void emit(uint32_t key, uint32_t outer_val, uint32_t inner_val)
    {
        if (!ok) return;                                   // aggregate-only mode (uniform)
        const u64 m = __ballot(1);                         // lanes active here = lanes with a match
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                              __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const u64 o = *cursor;                             // same address in every lane: broadcast
        const u64 room = (o == HJ_NO_CURSOR) ? 0 : ((o & ~(block_size - 1)) + block_size - o);
        u64 pos, next;
        if (n < room) {
            pos = o + rank;
            next = o + n;
        } else {
            // the run fills the current block: claim the next one (one atomic per block)
            u64 nb = 0;
            if (rank == 0) nb = atomicAdd(block_counter, 1ull);
            // broadcast from the first active lane (= the rank-0 lane)
            nb = ((u64)__builtin_amdgcn_readfirstlane((uint32_t)(nb >> 32)) << 32) |
                 (u64)__builtin_amdgcn_readfirstlane((uint32_t)nb);
            if (nb >= block_limit) {
                if (rank == 0) atomicOr(overflow, 1u);
                nb = block_limit - 1;                      // stay inside the allocation; result is flagged
            }
            const u64 base = nb * block_size;
            pos = (rank < room) ? (o + rank) : (base + (rank - room));
            next = base + (n - room);
        }
        if (rank == 0) *cursor = next;
        // Plain row stores that sit dirty in an XCD's L2 while another queue's kernel boundary writes back and invalidates it can
        // be lost (round 5, DESIGN section 3: K6 lost stores that way in 1.5 of 10^4 pipeline steps), and a materialising join runs
        // beside other streams' work in every pipeline (host batches, multi-GPU slices): non-temporal there.  They cost the rows
        // 14 % (4.83 against 4.24 ms per 10^9 rows: 4-byte stores do not fill lines the way K6's 16-byte ones do), which a solo join
        // need not pay.  HJ_ROW_STORE = 0 builds every instance with plain row stores (A/B).
        if constexpr (NT && HJ_ROW_STORE) {
            __builtin_nontemporal_store(key, &ok[pos]);
            __builtin_nontemporal_store(outer_val, &oov[pos]);
            __builtin_nontemporal_store(inner_val, &oiv[pos]);
        } else {
            ok[pos] = key;
            oov[pos] = outer_val;
            oiv[pos] = inner_val;
        }
    }

    // FOUR rows per lane at once: the lanes that call hold the four matches of one probe vector (one match per probe tuple: the common
    // case).  Every lane's rows are consecutive, so a column leaves the lane as ONE 16-byte store and the wave as one contiguous piece
    // of 64 x 16 bytes: whole 128-byte lines wherever the cursor stands on one.  Round 5 measured the 4-byte non-temporal row stores
    // at +14 % (4.83 against 4.24 ms per 10^9 rows) while K6's whole lines leave FASTER non-temporal than plain.
    // Needs block_size >= 512: the up to 256 rows of one call spill into at most ONE newly claimed block and never fill it (a cursor
    // must not come to rest on the first row of a block nobody claimed).
    __device__ __forceinline__ void emit4(const uint32_t (&key)[4], const uint32_t (&outer_val)[4], const uint32_t (&inner_val)[4])
    {
        if (!ok) return;
        const u64 m = __ballot(1);
        const uint32_t n = 4u * (uint32_t)__popcll(m);
        const uint32_t rank = 4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const u64 o = *cursor;
        const u64 room = (o == HJ_NO_CURSOR) ? 0 : ((o & ~(block_size - 1)) + block_size - o);
        u64 pos0, pos3, next, base = 0;
        if (n < room) {
            pos0 = o + rank; pos3 = pos0 + 3;
            next = o + n;
        } else {
            u64 nb = 0;
            if (rank == 0) nb = atomicAdd(block_counter, 1ull);
            nb = ((u64)__builtin_amdgcn_readfirstlane((uint32_t)(nb >> 32)) << 32) | (u64)__builtin_amdgcn_readfirstlane((uint32_t)nb);
            if (nb >= block_limit) {
                if (rank == 0) atomicOr(overflow, 1u);
                nb = block_limit - 1;
            }
            base = nb * block_size;
            pos0 = (rank < room) ? (o + rank) : (base + (rank - room));
            pos3 = (rank + 3 < room) ? (o + rank + 3) : (base + (rank + 3 - room));
            next = base + (n - room);
        }
        if (rank == 0) *cursor = next;
        if (pos3 == pos0 + 3) {
            typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
            const v4u_t k4 = {key[0], key[1], key[2], key[3]}, o4 = {outer_val[0], outer_val[1], outer_val[2], outer_val[3]},
                        i4 = {inner_val[0], inner_val[1], inner_val[2], inner_val[3]};
            if constexpr (NT && HJ_ROW_STORE) {
                __builtin_nontemporal_store(k4, reinterpret_cast<v4u_t *>(&ok[pos0]));
                __builtin_nontemporal_store(o4, reinterpret_cast<v4u_t *>(&oov[pos0]));
                __builtin_nontemporal_store(i4, reinterpret_cast<v4u_t *>(&oiv[pos0]));
            } else {
                *reinterpret_cast<v4u_t *>(&ok[pos0]) = k4;
                *reinterpret_cast<v4u_t *>(&oov[pos0]) = o4;
                *reinterpret_cast<v4u_t *>(&oiv[pos0]) = i4;
            }
        } else {
            // the lane's four rows straddle the end of the block: row by row (rows [0, room - rank) in the old block)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u64 pos = (rank + i < room) ? (o + rank + i) : (base + (rank + i - room));
                if constexpr (NT && HJ_ROW_STORE) {
                    __builtin_nontemporal_store(key[i], &ok[pos]); __builtin_nontemporal_store(outer_val[i], &oov[pos]); __builtin_nontemporal_store(inner_val[i], &oiv[pos]);
                } else { ok[pos] = key[i]; oov[pos] = outer_val[i]; oiv[pos] = inner_val[i]; }
            }
        }
    }
};
typedef EmitterT<true> Emitter;     // NPJ: always non-temporal rows
