// ubench_scatter_align.hip — does the ALIGNMENT of scattered runs matter on MI355X?
//
// Emulates the memory behaviour of K6 pass 1 without the sort: every workgroup streams
// tiles of 16384 packed tuples (128 KiB, 16-byte loads) and writes each tile as F runs of
// L = 16384/F tuples to F private frontiers (one region per (workgroup, partition), runs
// appended tile after tile).  `mis` shifts every frontier by that many 8-byte tuples, so
// mis = 0 gives runs that start on 128-byte lines when L*8 is a multiple of 128.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_scatter_align.hip -o hash_join_codes_knl_amd/lib/ubench_scatter_align
//   ./ubench_scatter_align [tuples=1<<30]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned long long u64;
constexpr int BLOCK = 1024, VPT = 8, TILE = BLOCK * VPT * 2;   // 2 tuples per 16-byte vector
template <int V> struct TileOf { static constexpr int value = BLOCK * V * 2; };

__global__ __launch_bounds__(BLOCK) void scatter_like(const uint4 *__restrict__ in, u64 *__restrict__ out,
                                                      u64 tiles, uint32_t F, uint32_t L, uint32_t mis)
{
    const u64 tiles_per_wg = (tiles + gridDim.x - 1) / gridDim.x;
    const u64 t0 = blockIdx.x * tiles_per_wg, t1 = min(tiles, t0 + tiles_per_wg);
    // frontier of (wg, p): region of tiles_per_wg * L tuples
    for (u64 t = t0; t < t1; ++t) {
        uint4 v[VPT];
#pragma unroll
        for (int j = 0; j < VPT; ++j) v[j] = in[t * (TILE / 2) + (u64)j * BLOCK + threadIdx.x];
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            // lane-contiguous stream-out: thread handles tuples 2*i and 2*i+1 of the "sorted" tile
            const uint32_t i = (j * BLOCK + threadIdx.x) * 2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint32_t pos = i + c;
                const uint32_t p = pos / L, r = pos - p * L;
                if (p < F) {
                    const u64 dst = ((u64)p * gridDim.x + blockIdx.x) * (tiles_per_wg * L + 16) + (t - t0) * L + r + mis;
                    out[dst] = c ? ((u64)v[j].w << 32 | v[j].z) : ((u64)v[j].y << 32 | v[j].x);
                }
            }
        }
    }
}

// 8-byte stores, one tuple per lane, consecutive lanes consecutive tuples (what K6 does)
// split > 0: the first `split` tuples of every run are written by an EARLIER instruction (a carry
//            flush), the rest of their line a barrier later: is a line completed microseconds later
//            by another store as good as a whole-line store?
// rot > 0:   lane -> tuple mapping rotated by `rot`, so every 4th line straddles two waves' stores
template <int VPT>
__global__ __launch_bounds__(BLOCK) void scatter_like8(const uint4 *__restrict__ in, u64 *__restrict__ out,
                                                       u64 tiles, uint32_t F, uint32_t L, uint32_t mis,
                                                       uint32_t split, uint32_t rot)
{
    constexpr int TILE = BLOCK * VPT * 2;
    extern __shared__ u64 stage[];
    const u64 tiles_per_wg = (tiles + gridDim.x - 1) / gridDim.x;
    const u64 t0 = blockIdx.x * tiles_per_wg, t1 = min(tiles, t0 + tiles_per_wg);
    for (u64 t = t0; t < t1; ++t) {
        uint4 v[VPT];
#pragma unroll
        for (int j = 0; j < VPT; ++j) v[j] = in[t * (TILE / 2) + (u64)j * BLOCK + threadIdx.x];
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const uint32_t i = (j * BLOCK + threadIdx.x) * 2;
            stage[i] = (u64)v[j].y << 32 | v[j].x;
            stage[i + 1] = (u64)v[j].w << 32 | v[j].z;
        }
        __syncthreads();
        if (split) {
            for (uint32_t idx = threadIdx.x; idx < F * 16; idx += BLOCK) {
                const uint32_t p = idx >> 4, r = idx & 15;
                if (r < split) {
                    const u64 dst = ((u64)p * gridDim.x + blockIdx.x) * (tiles_per_wg * L + 16) + (t - t0) * L + r + mis;
                    out[dst] = stage[p * L + r];
                }
            }
            __syncthreads();
        }
        for (uint32_t q = threadIdx.x; q < (uint32_t)TILE; q += BLOCK) {
            const uint32_t pos = rot ? (q + rot) % (uint32_t)TILE : q;
            const uint32_t p = pos / L, r = pos - p * L;
            if (p < F && r >= split) {
                const u64 dst = ((u64)p * gridDim.x + blockIdx.x) * (tiles_per_wg * L + 16) + (t - t0) * L + r + mis;
                out[dst] = stage[pos];
            }
        }
        __syncthreads();
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const u64 n = argc > 1 ? strtoull(argv[1], 0, 0) : (1ull << 30);
    uint4 *in; u64 *out;
    CK(hipMalloc(&in, n * 8 + 4096));
    CK(hipMalloc(&out, n * 8 + (1ull << 30)));
    CK(hipMemset(in, 1, n * 8));
    CK(hipMemset(out, 0, n * 8 + (1ull << 30)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&scatter_like8<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&scatter_like8<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
    struct Case { uint32_t F, L, mis; int k8; uint32_t split, rot; };   // k8: 0 = direct 16K tile, 8 = LDS 16K tile, 4 = LDS 8K tile
    const Case cases[] = {
        {128, 128, 0, 8, 0, 0}, {128, 128, 5, 8, 0, 0}, {128, 128, 0, 8, 5, 0}, {128, 128, 0, 8, 0, 5}, {128, 128, 0, 8, 0, 16},
        {512, 16, 0, 4, 0, 0}, {512, 16, 0, 4, 5, 0}, {512, 16, 0, 4, 0, 5},
    };
    for (const Case &c : cases) {
        const int tile = c.k8 == 4 ? 8192 : 16384;
        const u64 tiles = n / tile;
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(a, 0));
            if (c.k8 == 8) hipLaunchKernelGGL(scatter_like8<8>, dim3(grid), dim3(BLOCK), 16384 * 8, 0, in, out, tiles, c.F, c.L, c.mis, c.split, c.rot);
            else if (c.k8 == 4) hipLaunchKernelGGL(scatter_like8<4>, dim3(grid), dim3(BLOCK), 8192 * 8, 0, in, out, tiles, c.F, c.L, c.mis, c.split, c.rot);
            else hipLaunchKernelGGL(scatter_like, dim3(grid), dim3(BLOCK), 0, 0, in, out, tiles, c.F, c.L, c.mis);
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best) best = ms;
        }
        const double bytes = (double)tiles * tile * 8 + (double)tiles * c.F * c.L * 8;
        printf("%s tile=%5d F=%4u L=%5u mis=%2u split=%u rot=%2u : %.3f ms  %.0f GB/s (r+w)\n", c.k8 ? "lds+8B " : "direct ", tile, c.F, c.L, c.mis, c.split, c.rot,
               best, bytes / best / 1e6);
        fflush(stdout);
    }
    return 0;
}
