// hj_device.hpp — device-side helpers shared by the gfx950 kernels.
// wave = 64 lanes everywhere (CDNA4); no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;

// H(key, f, N) = mulhi32((uint32)(key*f), N)   (npj.cpp:200-201, phj.cpp:721-722)
__device__ __forceinline__ uint32_t hj_hash(uint32_t key, uint32_t factor, uint32_t n)
{
    return __umulhi(key * factor, n);
}

// Final partition id of a key under the two partitioning passes.
__device__ __forceinline__ uint32_t hj_part2(uint32_t key, uint32_t f1, uint32_t F1,
                                             uint32_t f2, uint32_t F2)
{
    return hj_hash(key, f1, F1) * F2 + hj_hash(key, f2, F2);
}

__device__ __forceinline__ int hj_lane() { return threadIdx.x & 63; }

// 16-byte non-temporal load (global_load_dwordx4 ... nt): for streams that are read exactly once
__device__ __forceinline__ uint4 hj_load_nt(const uint4 *p)
{
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    const v4u_t t = __builtin_nontemporal_load(reinterpret_cast<const v4u_t *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

// THE STORE POLICY (round 6).  K6's plain stores were LOST IN MEMORY beside other queues' kernels - 1.3-1.5 in 10^4 steps of the
// multi-stream pipelines, the damaged partitions read back wrong through hipMemcpy and through a fresh kernel with the device quiet
// (profiles/r06_lost_or_stale.txt) - and never with non-temporal stores.  The mechanism is below the ISA and a stand-alone kernel pair
// does not show it (tools/ubench_partial_line_race.hip), so nothing says that other kernels' plain stores are exempt: EVERY global
// store of every kernel of the library is non-temporal (hj_store), the two solo forms excepted (option "solo": K6's partial-line stores
// and the result rows of a blocking join that runs alone on the device).  tests/test_store_policy_isa.py reads the machine code of
// every kernel source and fails on a plain global store outside that list.  Atomics are performed at the memory side.
__device__ __forceinline__ void hj_store(uint32_t *p, uint32_t v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void hj_store(u64 *p, u64 v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void hj_store(uint4 *p, uint4 v)
{
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    const v4u_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4u_t *>(p));
}

// Workgroup barrier that orders LDS traffic only.  HIP's __syncthreads() also drains
// the vector-memory counter (s_waitcnt vmcnt(0)), which turns every global load issued
// before it into an exposed round trip; kernels that keep loads (or stores) in flight
// across a barrier use this one: wait for this wave's LDS operations, then s_barrier.
__device__ __forceinline__ void hj_barrier_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Inclusive scan across the 64 lanes of a wave.
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T x)
{
    if constexpr (sizeof(T) == 4) {
        // DPP: four row_shr steps inside the rows of 16 lanes (lanes without a source read 0), then the row totals
        // travel with row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3) - six VALU instructions instead
        // of six ds_bpermute round trips through the LDS crossbar (K6 scans its bins once per tile)
        uint32_t v = (uint32_t)x;
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);      // row_shr:1
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);      // row_shr:2
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);      // row_shr:4
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);      // row_shr:8
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
        return (T)v;
    } else {
        const int lane = hj_lane();
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            T y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        return x;
    }
}

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_down(x, d, 64);
    return x;   // valid in lane 0
}

// A value every lane of the wave holds identically (read from LDS, say), moved to scalar registers: what is
// computed from it - addresses of descriptor tables, loop bounds - becomes scalar work and scalar loads.
__device__ __forceinline__ uint32_t hj_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ u64 hj_uniform(u64 x)
{
    return ((u64)hj_uniform((uint32_t)(x >> 32)) << 32) | hj_uniform((uint32_t)x);
}

__device__ __forceinline__ uint32_t wave_reduce_max(uint32_t x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = max(x, (uint32_t)__shfl_down(x, d, 64));
    return x;   // valid in lane 0
}

// Block-wide exclusive scan of one value per thread. `scratch` needs
// BLOCK/64 + 1 entries of T in LDS; scratch[BLOCK/64] receives the block total.
// Contains two __syncthreads(); every thread of the block must call it.
// LDS_ONLY: ONE hj_barrier_lds() (global loads / stores / atomics stay in flight across it); scratch[BLOCK/64]
// is then only valid after the caller's next barrier, and scratch[0..NW) must not be rewritten before it.
template <int BLOCK, typename T, bool LDS_ONLY = false>
__device__ __forceinline__ T block_exclusive_scan(T v, T *scratch)
{
    constexpr int NW = BLOCK / 64;
    const int lane = hj_lane();
    const int wave = threadIdx.x >> 6;
    T inc = wave_inclusive_scan(v);
    if (lane == 63) scratch[wave] = inc;
    if (LDS_ONLY) {
        // one barrier: every thread adds up the totals of the waves before its own (NW broadcast reads);
        // thread 0 leaves the block total in scratch[NW] for whoever reads it after the caller's next barrier
        hj_barrier_lds();
        T before = T(0), total = T(0);
        if constexpr (sizeof(T) == 4 && NW % 4 == 0) {
            // the wave totals in 16-byte reads (4 instead of 16 LDS instructions per thread and call)
            const uint4 *s4 = reinterpret_cast<const uint4 *>(scratch);
#pragma unroll
            for (int q = 0; q < NW / 4; ++q) {
                const uint4 v = s4[q];
                const T x[4] = {T(v.x), T(v.y), T(v.z), T(v.w)};
#pragma unroll
                for (int e = 0; e < 4; ++e) { total += x[e]; if (q * 4 + e < wave) before += x[e]; }
            }
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) { const T x = scratch[w]; total += x; if (w < wave) before += x; }
        }
        if (threadIdx.x == 0) scratch[NW] = total;
        return inc - v + before;
    }
    __syncthreads();
    if (wave == 0) {
        T w = (lane < NW) ? scratch[lane] : T(0);
        T winc = wave_inclusive_scan(w);
        if (lane < NW) scratch[lane] = winc - w;      // exclusive prefix of the wave sums
        if (lane == NW - 1) scratch[NW] = winc;       // block total
    }
    __syncthreads();
    return inc - v + scratch[wave];
}
