// ubench_placement.hip — K6 pass 1 takes 2.9 or 3.4 ms on the same data depending on WHICH allocation its 8 GB output
// twin lives in (tools/alloc_luck.py, profiles/r02_alloc_luck.txt).  Is that visible to a cheap synthetic probe, so
// that a workspace allocator could pick a good placement?  For a series of freshly allocated 8.5 GB buffers this
// prints the time of (a) the write pattern of pass 1 alone: 256 workgroups x 128 partition frontiers, 1 KiB runs,
// ranges of 15 tiles claimed in order; (b) the same with the tile reads (the whole pass without its sort);
// (c) a plain streaming write of the buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_placement.hip -o hash_join_codes_knl_amd/lib/ubench_placement
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
constexpr int BLOCK = 1024, F = 128, TILE_BYTES = 128 * 1024, RUN = TILE_BYTES / F;   // 1 KiB per partition and tile

template <bool READ>
__global__ __launch_bounds__(BLOCK) void scatter_like(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 tiles,
                                                      u64 tiles_per_range, unsigned *ticket)
{
    __shared__ unsigned range_s;
    const u64 ranges = (tiles + tiles_per_range - 1) / tiles_per_range;
    const u64 region = (tiles * RUN + 127) / 128 * 128;                 // bytes per partition
    for (;;) {
        if (threadIdx.x == 0) range_s = atomicAdd(ticket, 1u);
        __syncthreads();
        const u64 r = range_s;
        __syncthreads();
        if (r >= ranges) return;
        for (u64 t = r * tiles_per_range; t < (r + 1) * tiles_per_range && t < tiles; ++t) {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = READ ? in[t * (TILE_BYTES / 16) + j * BLOCK + threadIdx.x] : make_uint4(j, threadIdx.x, 2, 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned s = j * BLOCK + threadIdx.x, p = s / (RUN / 16), w = s % (RUN / 16);
                out[(p * region + t * RUN) / 16 + w] = v[j];
            }
        }
    }
}

__global__ __launch_bounds__(BLOCK) void fill(uint4 *__restrict__ out, u64 n)
{
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) out[i] = make_uint4(1, 2, 3, 4);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10;
    const bool stacked = argc > 2 && atoi(argv[2]);
    const bool contiguous = argc > 3 && atoi(argv[3]) == 1;     // hipExtMallocWithFlags(hipDeviceMallocContiguous): physically contiguous
    // 2 / 3: one virtual range stitched from physical chunks of <chunk MiB> (argv[4], default 32) with the virtual-memory API,
    // the chunks mapped in creation order (2) or in a shuffled order (3): does the ORDER of a block's pages decide its kind?
    const int vmm = argc > 3 ? (atoi(argv[3]) >= 2 ? atoi(argv[3]) : 0) : 0;
    const size_t chunk_mib = argc > 4 ? (size_t)atoi(argv[4]) : 32;
    const u64 bytes = 8512ull * 1000 * 1000 / TILE_BYTES * TILE_BYTES, tiles = bytes / TILE_BYTES;
    uint4 *in; unsigned *ticket;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&ticket, 4)); CK(hipMemset(in, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](auto fn) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(ticket, 0, 4, 0));
            CK(hipEventRecord(e0, 0)); fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return best;
    };
    void *keep[64]; int kept = 0;
    for (int i = 0; i < n; ++i) {
        uint4 *out;
        if (vmm) {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0;
            CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            size_t chunk = chunk_mib << 20;
            chunk = (chunk + gran - 1) / gran * gran;
            const size_t total = (bytes + F * 256 + chunk - 1) / chunk * chunk, nchunks = total / chunk;
            void *va = nullptr;
            CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
            size_t *order = (size_t *)malloc(nchunks * sizeof(size_t));
            for (size_t k = 0; k < nchunks; ++k) order[k] = k;
            if (vmm == 3) { unsigned long long x = 88172645463325252ull + i; for (size_t k = nchunks - 1; k > 0; --k) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; size_t j = x % (k + 1); size_t t = order[k]; order[k] = order[j]; order[j] = t; } }
            hipMemGenericAllocationHandle_t *h = (hipMemGenericAllocationHandle_t *)malloc(nchunks * sizeof(*h));
            for (size_t k = 0; k < nchunks; ++k) CK(hipMemCreate(&h[k], chunk, &prop, 0));
            for (size_t k = 0; k < nchunks; ++k) CK(hipMemMap((char *)va + k * chunk, chunk, 0, h[order[k]], 0));
            hipMemAccessDesc acc = {};
            acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, total, &acc, 1));
            out = (uint4 *)va;
            free(order); free(h);                                     // (the handles and the range live until the process ends)
        } else if (contiguous) {
            if (hipExtMallocWithFlags((void **)&out, bytes + F * 256, hipDeviceMallocContiguous) != hipSuccess) { printf("contiguous allocation %d refused\n", i); (void)hipGetLastError(); break; }
        } else CK(hipMalloc(&out, bytes + F * 256));
        const float w = timed([&] { hipLaunchKernelGGL(scatter_like<false>, dim3(256), dim3(BLOCK), 0, 0, in, out, tiles, 15, ticket); });
        const float rw = timed([&] { hipLaunchKernelGGL(scatter_like<true>, dim3(256), dim3(BLOCK), 0, 0, in, out, tiles, 15, ticket); });
        const float f = timed([&] { hipLaunchKernelGGL(fill, dim3(1024), dim3(BLOCK), 0, 0, out, bytes / 16); });
        printf("%sallocation %2d at %p: scattered writes %.3f ms, read + scattered writes %.3f ms, streaming fill %.3f ms\n", vmm == 3 ? "stitched, shuffled " : vmm == 2 ? "stitched, in order " : contiguous ? "contiguous " : "", i, (void *)out, w, rw, f);
        fflush(stdout);
        if (vmm) continue;                                            // stays mapped
        if (stacked && kept < 8) keep[kept++] = out; else CK(hipFree(out));
    }
    return 0;
}
