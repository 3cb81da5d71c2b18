// ubench_copy_sweep.hip — what does a plain copy reach on this MI355X, and with which shape?
//
// MI355X_MICROARCH.md quotes 6.29 TB/s for a "float4 copy" (read + written bytes / time); round 1's
// persistent-grid copies (tools/ubench_copy.hip) and hipMemcpyAsync reached 4.7-5.2 TB/s.  This sweep
// covers the shapes round 1 did not try, so that K6's ceiling is a measurement and not a claim:
//   one-shot grids (one workgroup per U x block vectors, no loop),
//   persistent grids with a grid-stride loop (round 1's shape) at more unroll depths,
//   blocked copies (a workgroup owns contiguous chunks of 64 KiB .. 4 MiB),
//   non-temporal loads and / or stores, 256 / 512 / 1024-thread workgroups, 1 GiB and 8 GiB buffers,
//   separate source/destination allocations vs the two halves of one allocation.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_copy_sweep.hip -o hash_join_codes_knl_amd/lib/ubench_copy_sweep
//   ./ubench_copy_sweep [GiB=8]          prints one line per variant, best of 5 after a warm-up
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld(const uint4 *p)
{
    if (NT) { const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(uint4 *p, uint4 v)
{
    if (NT) { v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(p)); }
    else *p = v;
}

// one-shot: workgroup b moves vectors [b * BLOCK * U, (b + 1) * BLOCK * U), lane-interleaved
template <int BLOCK, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void oneshot_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n)
{
    const u64 base = (u64)blockIdx.x * BLOCK * U + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) if (base + (u64)j * BLOCK < n) v[j] = ld<NTL>(in + base + (u64)j * BLOCK);
#pragma unroll
    for (int j = 0; j < U; ++j) if (base + (u64)j * BLOCK < n) st<NTS>(out + base + (u64)j * BLOCK, v[j]);
}

// persistent, grid-stride: the whole chip sweeps memory together
template <int BLOCK, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void stride_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n)
{
    const u64 step = (u64)gridDim.x * BLOCK * U;
    for (u64 base = (u64)blockIdx.x * BLOCK * U + threadIdx.x; base < n; base += step) {
        uint4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) if (base + (u64)j * BLOCK < n) v[j] = ld<NTL>(in + base + (u64)j * BLOCK);
#pragma unroll
        for (int j = 0; j < U; ++j) if (base + (u64)j * BLOCK < n) st<NTS>(out + base + (u64)j * BLOCK, v[j]);
    }
}

// persistent, blocked: a workgroup owns contiguous chunks of `chunk` vectors (claimed round-robin)
template <int BLOCK, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void blocked_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n, u64 chunk)
{
    const u64 chunks = (n + chunk - 1) / chunk;
    for (u64 c = blockIdx.x; c < chunks; c += gridDim.x) {
        const u64 lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
        for (u64 base = lo + threadIdx.x; base < hi; base += (u64)BLOCK * U) {
            uint4 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) if (base + (u64)j * BLOCK < hi) v[j] = ld<NTL>(in + base + (u64)j * BLOCK);
#pragma unroll
            for (int j = 0; j < U; ++j) if (base + (u64)j * BLOCK < hi) st<NTS>(out + base + (u64)j * BLOCK, v[j]);
        }
    }
}

// read-only and write-only references in the best shapes (for the r : w split of a copy)
template <int BLOCK, int U, bool NTL>
__global__ __launch_bounds__(BLOCK) void read_kernel(const uint4 *__restrict__ in, u64 n, uint4 *sink)
{
    const u64 step = (u64)gridDim.x * BLOCK * U;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u64 base = (u64)blockIdx.x * BLOCK * U + threadIdx.x; base < n; base += step) {
#pragma unroll
        for (int j = 0; j < U; ++j) if (base + (u64)j * BLOCK < n) { const uint4 v = ld<NTL>(in + base + (u64)j * BLOCK); acc.x ^= v.x; acc.y ^= v.w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) *sink = acc;
}
template <int BLOCK, int U, bool NTS>
__global__ __launch_bounds__(BLOCK) void write_kernel(uint4 *__restrict__ out, u64 n)
{
    const u64 step = (u64)gridDim.x * BLOCK * U;
    const uint4 v = make_uint4(threadIdx.x, 1, 2, 3);
    for (u64 base = (u64)blockIdx.x * BLOCK * U + threadIdx.x; base < n; base += step) {
#pragma unroll
        for (int j = 0; j < U; ++j) if (base + (u64)j * BLOCK < n) st<NTS>(out + base + (u64)j * BLOCK, v);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static hipEvent_t ev_a, ev_b;
template <typename F>
static float best_ms(F launch)
{
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(ev_a, 0));
        launch();
        CK(hipEventRecord(ev_b, 0));
        CK(hipEventSynchronize(ev_b));
        float ms; CK(hipEventElapsedTime(&ms, ev_a, ev_b));
        if (rep && ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

static void report(const char *what, double bytes_moved, float ms)
{
    printf("%-72s %8.3f ms  %6.0f GB/s\n", what, ms, bytes_moved / ms / 1e6);
    fflush(stdout);
}

template <int BLOCK, int U, bool NTL, bool NTS>
static void run_shapes(const uint4 *in, uint4 *out, u64 n, int cus, const char *tag)
{
    char buf[160];
    const double moved = 2.0 * 16.0 * (double)n;
    const u64 per = (u64)BLOCK * U;
    snprintf(buf, sizeof buf, "%s copy one-shot      block %4d U %d ntl %d nts %d", tag, BLOCK, U, NTL, NTS);
    report(buf, moved, best_ms([&] { hipLaunchKernelGGL((oneshot_kernel<BLOCK, U, NTL, NTS>), dim3((unsigned)((n + per - 1) / per)), dim3(BLOCK), 0, 0, in, out, n); }));
    for (int k : {1, 2, 4, 8}) {
        if (k * BLOCK > 2048) continue;
        snprintf(buf, sizeof buf, "%s copy grid-stride   block %4d U %d ntl %d nts %d grid %d x CUs", tag, BLOCK, U, NTL, NTS, k);
        report(buf, moved, best_ms([&] { hipLaunchKernelGGL((stride_kernel<BLOCK, U, NTL, NTS>), dim3(cus * k), dim3(BLOCK), 0, 0, in, out, n); }));
    }
    for (u64 chunk_kib : {64ull, 512ull, 4096ull}) {
        const int k = 2048 / BLOCK;
        snprintf(buf, sizeof buf, "%s copy blocked %4llu KiB block %4d U %d ntl %d nts %d grid %d x CUs", tag, chunk_kib, BLOCK, U, NTL, NTS, k);
        report(buf, moved, best_ms([&] { hipLaunchKernelGGL((blocked_kernel<BLOCK, U, NTL, NTS>), dim3(cus * k), dim3(BLOCK), 0, 0, in, out, n, chunk_kib * 64); }));
    }
}

int main(int argc, char **argv)
{
    const u64 gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, memory clock %d kHz, bus %d bits\n", prop.name, cus, prop.memoryClockRate, prop.memoryBusWidth);
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    uint4 *sink; CK(hipMalloc(&sink, 16));
    for (u64 g : {gib, (u64)1}) {
        const u64 bytes = g << 30, n = bytes / 16;
        uint4 *in, *out, *both;
        CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&both, 2 * bytes));
        CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes)); CK(hipMemset(both, 1, 2 * bytes));
        char tag[32];
        snprintf(tag, sizeof tag, "%2llu GiB", g);
        char buf[160];
        // the runtime's own copy and fill
        snprintf(buf, sizeof buf, "%s hipMemcpyAsync device to device", tag);
        report(buf, 2.0 * bytes, best_ms([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }));
        snprintf(buf, sizeof buf, "%s hipMemsetAsync", tag);
        report(buf, 1.0 * bytes, best_ms([&] { CK(hipMemsetAsync(out, 0, bytes, 0)); }));
        // reads and writes alone
        snprintf(buf, sizeof buf, "%s read  grid-stride block 256 U 4 grid 8 x CUs", tag);
        report(buf, 1.0 * bytes, best_ms([&] { hipLaunchKernelGGL((read_kernel<256, 4, false>), dim3(cus * 8), dim3(256), 0, 0, in, n, sink); }));
        snprintf(buf, sizeof buf, "%s read  grid-stride block 256 U 8 nt grid 8 x CUs", tag);
        report(buf, 1.0 * bytes, best_ms([&] { hipLaunchKernelGGL((read_kernel<256, 8, true>), dim3(cus * 8), dim3(256), 0, 0, in, n, sink); }));
        snprintf(buf, sizeof buf, "%s write grid-stride block 256 U 4 grid 8 x CUs", tag);
        report(buf, 1.0 * bytes, best_ms([&] { hipLaunchKernelGGL((write_kernel<256, 4, false>), dim3(cus * 8), dim3(256), 0, 0, out, n); }));
        snprintf(buf, sizeof buf, "%s write grid-stride block 256 U 4 nt grid 8 x CUs", tag);
        report(buf, 1.0 * bytes, best_ms([&] { hipLaunchKernelGGL((write_kernel<256, 4, true>), dim3(cus * 8), dim3(256), 0, 0, out, n); }));
        // copies
        run_shapes<256, 1, false, false>(in, out, n, cus, tag);
        run_shapes<256, 4, false, false>(in, out, n, cus, tag);
        run_shapes<256, 8, false, false>(in, out, n, cus, tag);
        run_shapes<512, 4, false, false>(in, out, n, cus, tag);
        run_shapes<1024, 4, false, false>(in, out, n, cus, tag);
        run_shapes<1024, 8, false, false>(in, out, n, cus, tag);
        run_shapes<256, 4, true, false>(in, out, n, cus, tag);
        run_shapes<256, 4, false, true>(in, out, n, cus, tag);
        run_shapes<256, 4, true, true>(in, out, n, cus, tag);
        run_shapes<1024, 4, true, true>(in, out, n, cus, tag);
        // source and destination as the two halves of ONE allocation (K6's twins are separate allocations)
        snprintf(buf, sizeof buf, "%s halves of one allocation:", tag);
        printf("%s\n", buf);
        run_shapes<256, 4, false, false>(both, both + n, n, cus, tag);
        CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(both));
    }
    return 0;
}
