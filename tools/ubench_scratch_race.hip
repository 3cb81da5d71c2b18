// ubench_scratch_race.hip — is a wave's scratch (private segment) safe while kernels of OTHER queues share the CUs?
// Round 3's multi-GPU stress runs lost tuples only with a pass-2 instance that spilled six VGPRs; the review asked for
// a reproducer without any join code.  Two kernels use scratch here, shaped like K6 pass 2 (1024 threads, 150 KiB of
// LDS = one workgroup per CU, persistent grid, work claimed from a ticket counter, a streaming read + LDS exchange +
// streaming write per tile):
//   explicit : a volatile private array (clang keeps it in the private segment: scratch_store / scratch_load), six
//              words written once before the tile loop and re-read every tile (the shape of a spill), six rewritten
//              and re-read every tile;
//   spill    : 96 live 32-bit values per thread, opaque to the compiler, at the 128-VGPR cap of a 1024-thread
//              workgroup: the register allocator spills some of them (-Rpass-analysis=kernel-resource-usage says how
//              many); all are verified every tile.
// Every re-read value is compared with its recomputed pattern; mismatches are counted and the first few recorded
// (thread, slot, expected, got).  Neighbours: a scratch-free persistent copy kernel with 150 KiB of LDS (K6 pass 1's
// shape) and one with 512 threads / 64 KiB (the join's shape), on streams of other priorities, plus device-to-device
// copies on a high-priority stream (the exchange).
// The checked kernels also STAMP their output (launch, tile, slot) and a verification kernel counts slots that kept an
// older stamp - stores that never reached memory (round 4: that is what a private segment in K6 pass 1 costs next to
// other queues' kernels; the private values themselves always come back right).
//   modes: 0 alone | 1 + big-LDS neighbour on a low-priority stream | 2 the slice pipeline's shape on three streams of three
//          priorities | 3 two checked kernels on two streams | 4 mode 2 + a second checked kernel | 5 the checked kernel on
//          its own stream next to a stream of big-LDS and join-shaped kernels, one priority | 6 mode 5, checked stream at low priority
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_scratch_race.hip -o hash_join_codes_knl_amd/lib/ubench_scratch_race
//   usage: ubench_scratch_race [seconds per mode = 4] [mode = all]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef unsigned long long u64;

constexpr int BLOCK = 1024;
constexpr int TILE_VEC = 4;                              // 16-byte vectors per thread and tile
constexpr u64 TILE_BYTES = (u64)BLOCK * TILE_VEC * 16;   // 64 KiB
#define L8(p) X(p##0) X(p##1) X(p##2) X(p##3) X(p##4) X(p##5) X(p##6) X(p##7)
#define LIVE_LIST L8(la) L8(lb) L8(lc) L8(ld) L8(le) L8(lf) L8(lg) L8(lh) L8(li) L8(lj) L8(lk) L8(ll)
constexpr int NLIVE = 96;

struct Report {
    u64 errors;
    u64 checks;
    u64 sample[8][4];     // thread, slot, expected, got
};

__device__ __forceinline__ uint32_t pattern(uint32_t gtid, uint32_t slot, uint32_t salt)
{
    uint32_t x = gtid * 0x9E3779B1u + slot * 0x85EBCA6Bu + salt * 0xC2B2AE35u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x | 1u;
}

__device__ __forceinline__ void complain(Report *rep, uint32_t gtid, uint32_t slot, uint32_t want, uint32_t got)
{
    const u64 at = atomicAdd(&rep->errors, 1ull);
    if (at < 8) { rep->sample[at][0] = gtid; rep->sample[at][1] = slot; rep->sample[at][2] = want; rep->sample[at][3] = got; }
}

// one tile of "work": streaming read, an exchange through LDS behind a barrier, streaming write
// K6's barrier: orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier), so global loads and STORES stay in flight
// across it - and across the end of the wave; __syncthreads() would drain the vector-memory counter every time.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// The output is stamped (salt of the launch, tile, slot): a slot that keeps an older launch's stamp was never written.
__device__ __forceinline__ void tile_work(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 tile, uint4 *lds, int lds_vecs,
                                          uint32_t salt = 0)
{
    const int tid = threadIdx.x;
    uint4 v[TILE_VEC];
#pragma unroll
    for (int j = 0; j < TILE_VEC; ++j) v[j] = in[tile * (TILE_BYTES / 16) + j * blockDim.x + tid];
#pragma unroll
    for (int j = 0; j < TILE_VEC; ++j) lds[(j * blockDim.x + tid) % lds_vecs] = v[j];
    lds_barrier();
#pragma unroll
    for (int j = 0; j < TILE_VEC; ++j) {
        const uint4 w = lds[(j * blockDim.x + (tid ^ 37)) % lds_vecs];
        out[tile * (TILE_BYTES / 16) + j * blockDim.x + tid] = make_uint4(salt, (uint32_t)tile, j * blockDim.x + tid, w.x ^ v[j].x);
    }
    lds_barrier();
}

template <bool SCRATCH, bool DRAIN>
__global__ __launch_bounds__(BLOCK) void scratch_explicit_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 tiles,
                                                                 unsigned *ticket, Report *rep, uint32_t salt, int lds_vecs)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem);
    __shared__ unsigned tile_s;
    const uint32_t gtid = blockIdx.x * BLOCK + threadIdx.x;
    volatile uint32_t priv[12];                            // the private segment: scratch (SCRATCH = false: the same kernel without it)
    if (SCRATCH) for (int j = 0; j < 12; ++j) priv[j] = pattern(gtid, j, salt);
    u64 checks = 0;
    for (;;) {
        if (threadIdx.x == 0) tile_s = atomicAdd(ticket, 1u);
        lds_barrier();
        const u64 t = tile_s;
        lds_barrier();
        if (t >= tiles) break;
        if (SCRATCH) for (int j = 6; j < 12; ++j) priv[j] = pattern(gtid, j, salt + (uint32_t)t);     // rewritten per tile
        tile_work(in, out, t, lds, lds_vecs, salt);
        if (SCRATCH) for (int j = 0; j < 12; ++j) {
            const uint32_t want = pattern(gtid, j, j < 6 ? salt : salt + (uint32_t)t), got = priv[j];
            if (got != want) complain(rep, gtid, j, want, got);
        }
        checks += SCRATCH ? 12 : 0;
    }
    if (threadIdx.x == 0) atomicAdd(&rep->checks, checks * BLOCK);
    if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the wave ends with no vector-memory operation in flight
}

// counts the slots of `out` that do not carry the stamp of launch `salt`
__global__ __launch_bounds__(BLOCK) void verify_output_kernel(const uint4 *__restrict__ out, u64 tiles, uint32_t salt, u64 *stale)
{
    u64 bad = 0;
    for (u64 t = blockIdx.x; t < tiles; t += gridDim.x)
        for (int j = 0; j < TILE_VEC; ++j) {
            const uint4 v = out[t * (TILE_BYTES / 16) + j * BLOCK + threadIdx.x];
            if (v.x != salt || v.y != (uint32_t)t || v.z != (uint32_t)(j * BLOCK + threadIdx.x)) ++bad;
        }
    if (bad) atomicAdd(stale, bad);
}

__global__ __launch_bounds__(BLOCK) void scratch_spill_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 tiles,
                                                              unsigned *ticket, Report *rep, uint32_t salt, int lds_vecs)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem);
    __shared__ unsigned tile_s;
    const uint32_t gtid = blockIdx.x * BLOCK + threadIdx.x;
    // NLIVE named scalars (an array became one wide vector value that the allocator spilled whole)
    const uint32_t base = pattern(gtid, 0, salt);
    uint32_t k = 0;
#define X(n) uint32_t n = base + (k++) * 0x9E3779B1u; asm volatile("" : "+v"(n));     /* opaque: must be kept */
    LIVE_LIST
#undef X
    u64 checks = 0;
    for (;;) {
        if (threadIdx.x == 0) tile_s = atomicAdd(ticket, 1u);
        lds_barrier();
        const u64 t = tile_s;
        lds_barrier();
        if (t >= tiles) break;
        tile_work(in, out, t, lds, lds_vecs, salt);
        uint32_t want = pattern(gtid, 0, salt), bad = 0;
        asm volatile("" : "+v"(want));                     // recomputed every tile, not kept from `base`
        k = 0;
#define X(n) bad |= n ^ (want + (k++) * 0x9E3779B1u); asm volatile("" : "+v"(n));      /* stays live across the back edge */
        LIVE_LIST
#undef X
        if (bad) complain(rep, gtid, 1000, want, bad);      // got = OR of (value ^ expected) over the live values
        checks += NLIVE;
    }
    if (threadIdx.x == 0) atomicAdd(&rep->checks, checks * BLOCK);
}

// scratch-free neighbours: the same tile work, other workgroup shapes
template <int THREADS>
__global__ __launch_bounds__(THREADS) void neighbour_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 tiles,
                                                            unsigned *ticket, int lds_vecs)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem);
    __shared__ unsigned tile_s;
    const u64 scale = BLOCK / THREADS;                     // tiles are TILE_BYTES of a 1024-thread workgroup: smaller groups take fractions
    for (;;) {
        if (threadIdx.x == 0) tile_s = atomicAdd(ticket, 1u);
        __syncthreads();
        const u64 t = tile_s;
        __syncthreads();
        if (t >= tiles * scale) break;
        const int tid = threadIdx.x;
        uint4 v[TILE_VEC];
#pragma unroll
        for (int j = 0; j < TILE_VEC; ++j) v[j] = in[t * (TILE_BYTES / 16 / scale) + j * THREADS + tid];
#pragma unroll
        for (int j = 0; j < TILE_VEC; ++j) lds[(j * THREADS + tid) % lds_vecs] = v[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TILE_VEC; ++j) out[t * (TILE_BYTES / 16 / scale) + j * THREADS + tid] = lds[(j * THREADS + (tid ^ 5)) % lds_vecs];
        __syncthreads();
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int rt = 0, drv = 0;
    (void)hipRuntimeGetVersion(&rt); (void)hipDriverGetVersion(&drv);
    printf("device %s, %d CUs, HIP runtime %d, driver %d\n", prop.name, prop.multiProcessorCount, rt, drv);
    const int cus = prop.multiProcessorCount;
    const size_t big_lds = 150 * 1024, join_lds = 64 * 1024;
    typedef void (*kern_t)(const uint4 *, uint4 *, u64, unsigned *, Report *, uint32_t, int);
    // flavours of the checked kernel: private array | register spills | private array + drained tail | no private segment (control)
    kern_t flavour[4] = {scratch_explicit_kernel<true, false>, scratch_spill_kernel, scratch_explicit_kernel<true, true>, scratch_explicit_kernel<false, false>};
    const char *flavour_name[4] = {"private array", "register spills", "private array, s_waitcnt vmcnt(0) before the wave ends", "NO private segment (control)"};
    for (kern_t k : flavour) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&neighbour_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&neighbour_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)join_lds));
    for (int f = 0; f < 4; ++f) {
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(flavour[f])));
        printf("flavour %d (%s): %d VGPRs, %zu bytes of private segment per lane\n", f, flavour_name[f], fa.numRegs, (size_t)fa.localSizeBytes);
    }

    const u64 bytes = 2ull << 30, tiles = bytes / TILE_BYTES;              // 2 GiB per buffer, 32768 tiles: ~1 ms per kernel
    uint4 *in[3], *out[3];
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&in[i], bytes)); CK(hipMalloc(&out[i], bytes)); CK(hipMemset(in[i], 0x5a + i, bytes)); CK(hipMemset(out[i], 0, bytes)); }
    unsigned *tickets; Report *rep; u64 *stale;
    CK(hipMalloc(&tickets, 64 * sizeof(unsigned)));
    CK(hipMalloc(&rep, sizeof(Report)));
    CK(hipMalloc(&stale, sizeof(u64)));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));                          // lo = lowest priority (largest number)
    hipStream_t s_main, s_low, s_high, s_other;
    CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_low, hipStreamNonBlocking, lo));
    CK(hipStreamCreateWithPriority(&s_high, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&s_other, hipStreamNonBlocking));
    const int big_vecs = (int)(big_lds / 16), join_vecs = (int)(join_lds / 16);

    uint32_t last_salt[3] = {0, 0, 0};                                      // stamp of the last checked launch into out[buf]
    auto checked_on = [&](int f, hipStream_t s, int buf, int slot, uint32_t salt) {
        CK(hipMemsetAsync(tickets + slot, 0, 4, s));
        hipLaunchKernelGGL(flavour[f], dim3(cus), dim3(BLOCK), big_lds, s, in[buf], out[buf], tiles, tickets + slot, rep, salt, big_vecs);
        last_salt[buf] = salt;
    };
    auto big_on = [&](hipStream_t s, int buf, int slot) {
        CK(hipMemsetAsync(tickets + slot, 0, 4, s));
        hipLaunchKernelGGL(neighbour_kernel<1024>, dim3(cus), dim3(1024), big_lds, s, in[buf], out[buf], tiles, tickets + slot, big_vecs);
    };
    auto join_on = [&](hipStream_t s, int buf, int slot) {
        CK(hipMemsetAsync(tickets + slot, 0, 4, s));
        hipLaunchKernelGGL(neighbour_kernel<512>, dim3(2 * cus), dim3(512), join_lds, s, in[buf], out[buf], tiles / 2, tickets + slot, join_vecs);
    };

    // modes: where the checked kernel runs and what runs beside it (the checked kernel always writes out[0]; in modes 5 / 6
    // it sits where pass 1 sits in the slice pipeline: on its own stream, next to a stream of join-shaped kernels)
    const char *names[] = {"alone", "checked kernel + big-LDS neighbour on a low-priority stream", "checked kernel, then join-shaped kernel on one stream; big-LDS neighbour (low priority); copies (high priority)",
                           "two checked kernels on two streams", "mode 2 with a second checked kernel on a fourth stream",
                           "checked kernel on its own stream next to a stream of big-LDS + join-shaped kernels (all default priority)",
                           "mode 5, the checked kernel's stream at low priority"};
    for (int mode = 0; mode < 7; ++mode) {
        if (only >= 0 && mode != only) continue;
        for (int f = 0; f < 4; ++f) {
            CK(hipMemset(rep, 0, sizeof(Report)));
            CK(hipMemset(stale, 0, sizeof(u64)));
            const auto t0 = std::chrono::steady_clock::now();
            u64 launches = 0, verified = 0;
            uint32_t salt = 1 + 1000003u * (uint32_t)(mode * 4 + f);
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds / 4) {
                for (int rnd = 0; rnd < 4; ++rnd, ++salt, ++launches) {
                    if (mode == 0) checked_on(f, s_main, 0, 0, salt);
                    if (mode == 1) { big_on(s_low, 1, 1); checked_on(f, s_main, 0, 0, salt); }
                    if (mode == 2 || mode == 4) {
                        big_on(s_low, 1, 1);
                        checked_on(f, s_main, 0, 0, salt);
                        if (mode == 4) checked_on(f, s_other, 2, 3, salt);
                        join_on(s_main, 1, 2);
                        CK(hipMemcpyAsync(out[1], in[2], bytes / 4, hipMemcpyDeviceToDevice, s_high));
                    }
                    if (mode == 3) { checked_on(f, s_main, 0, 0, salt); checked_on(f, s_other, 2, 3, salt); }
                    if (mode == 5 || mode == 6) {
                        checked_on(f, mode == 5 ? s_other : s_low, 0, 0, salt);
                        big_on(s_main, 1, 1); join_on(s_main, 1, 2); join_on(s_main, 1, 2);
                    }
                }
                CK(hipDeviceSynchronize());
                // every slot of out[0] (and out[2]) must carry the stamp of the last launch that wrote it
                hipLaunchKernelGGL(verify_output_kernel, dim3(1024), dim3(BLOCK), 0, s_main, out[0], tiles, last_salt[0], stale);
                if (mode == 3 || mode == 4) hipLaunchKernelGGL(verify_output_kernel, dim3(1024), dim3(BLOCK), 0, s_main, out[2], tiles, last_salt[2], stale);
                CK(hipDeviceSynchronize());
                ++verified;
            }
            Report h; u64 hs = 0;
            CK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
            CK(hipMemcpy(&hs, stale, sizeof(hs), hipMemcpyDeviceToHost));
            printf("mode %d, %s: %llu launches, %llu private values re-read, %llu WRONG; %llu outputs verified (%llu slots each), %llu STALE SLOTS\n",
                   mode, flavour_name[f], launches, h.checks, h.errors, verified, (u64)tiles * TILE_VEC * BLOCK, hs);
            for (u64 i = 0; i < h.errors && i < 8; ++i)
                printf("    thread %llu slot %llu: expected %08llx got %08llx\n", h.sample[i][0], h.sample[i][1], h.sample[i][2], h.sample[i][3]);
            fflush(stdout);
        }
        printf("  (mode %d = %s)\n", mode, names[mode]);
    }
    return 0;
}
