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

// The cursor is addressed as LDS (address space 3), not through a generic pointer: with a generic `volatile u64 *`
// the compiler emitted flat_load / flat_store with `s_waitcnt vmcnt(0)` around every emit - FLAT operations count on
// the vector-memory counter, so each emit first waited for EVERY outstanding global load and store of the wave
// (rounds 1-2: the materialising join ran at 0.56-0.59 of the peak on read + written bytes for that reason).
typedef __attribute__((address_space(3))) u64 hj_lds_u64;

struct Emitter {
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
#if defined(HJ_EXP_NOSTORE)          // tools/build_variant.py experiments (timing only, wrong rows): never defined in the product build
        (void)pos;
#elif defined(HJ_EXP_ONECOL)
        ok[pos] = key;
#else
        ok[pos] = key;
        oov[pos] = outer_val;
        oiv[pos] = inner_val;
#endif
    }

    // Four rows per lane in one go: called by a (sub)set of lanes that ALL have exactly one match for each of the four
    // keys of their probe vector (the common case with unique build keys and selectivity 1: every vector of the
    // stream).  Lane `rank` writes rows [o + 4 rank, o + 4 rank + 4) of the wave's run: one 16-byte store per column
    // and lane - a column leaves the wave as one 1 KiB piece per instruction instead of four 256-byte pieces -, one
    // cursor update and one ballot per 256 rows instead of four (npj.cpp:292-317 stages 256 entries and streams them
    // out the same way).  Returns false (uniformly, nothing written) when the run does not fit the wave's current
    // block: the caller then emits key by key, and that path claims the next block.
    // The stores are only 4-byte aligned when the cursor is not a multiple of 4 (after key-by-key emits): gfx950
    // executes dword-aligned global_store_dwordx4 (unaligned access mode), the type below tells the compiler so.
    typedef uint32_t row4_t __attribute__((ext_vector_type(4), aligned(4)));
    __device__ __forceinline__ bool emit4(const uint32_t (&key)[4], const uint32_t (&outer_val)[4],
                                          const uint32_t (&inner_val)[4])
    {
        const u64 m = __ballot(1);
        const uint32_t n = 4u * (uint32_t)__popcll(m);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const u64 o = *cursor;
        if (o == HJ_NO_CURSOR) return false;
        const u64 room = (o & ~(block_size - 1)) + block_size - o;
        if (n >= room) return false;
        const u64 pos = o + 4u * rank;
        if (rank == 0) *cursor = o + n;
        row4_t k4 = {key[0], key[1], key[2], key[3]};
        row4_t o4 = {outer_val[0], outer_val[1], outer_val[2], outer_val[3]};
        row4_t i4 = {inner_val[0], inner_val[1], inner_val[2], inner_val[3]};
#if defined(HJ_EXP_NOSTORE)
        (void)k4; (void)o4; (void)i4; (void)pos;
#elif defined(HJ_EXP_ONECOL)
        *reinterpret_cast<row4_t *>(ok + pos) = k4;
#elif defined(HJ_EXP_NTSTORE)
        __builtin_nontemporal_store(k4, reinterpret_cast<row4_t *>(ok + pos));
        __builtin_nontemporal_store(o4, reinterpret_cast<row4_t *>(oov + pos));
        __builtin_nontemporal_store(i4, reinterpret_cast<row4_t *>(oiv + pos));
#else
        *reinterpret_cast<row4_t *>(ok + pos) = k4;
        *reinterpret_cast<row4_t *>(oov + pos) = o4;
        *reinterpret_cast<row4_t *>(oiv + pos) = i4;
#endif
        return true;
    }
};
