// ubench_copy.hip — the box's streaming ceilings: 16-byte read-only, write-only and copy kernels over
// 8 GiB, persistent grids of 256..4096 workgroups.  K4/K7 are read streams, K6 is a copy with a detour.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o hash_join_codes_knl_amd/lib/ubench_copy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned long long u64;

template <int MODE>   // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void stream_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n, uint4 *sink)
{
    const u64 stride = (u64)gridDim.x * 256;
    uint4 acc = make_uint4(0, 0, 0, 0);
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        if (MODE == 1) { out[i] = acc; out[i + stride] = acc; out[i + 2 * stride] = acc; out[i + 3 * stride] = acc; continue; }
        const uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        if (MODE == 2) { out[i] = a; out[i + stride] = b; out[i + 2 * stride] = c; out[i + 3 * stride] = d; }
        else { acc.x += a.x ^ b.y; acc.y += c.z ^ d.w; }
    }
    if (MODE == 0 && acc.x == 0x12345678u) *sink = acc;
}

// K4-like read: 1024-thread workgroups, 64 KiB "tiles" (16 bytes per thread x 4), a workgroup owns
// RANGES of `k` tiles: consecutive tiles (contig = 1: every workgroup is its own sequential stream)
// or tiles interleaved with the other workgroups' (contig = 0: the chip sweeps memory together)
__global__ __launch_bounds__(1024) void range_read_kernel(const uint4 *__restrict__ in, u64 tiles, uint32_t k, int contig, uint4 *sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    const u64 G = gridDim.x, ranges = (tiles + k - 1) / k;
    for (u64 r = blockIdx.x; r < ranges; r += G) {
        for (uint32_t i = 0; i < k; ++i) {
            const u64 t = contig ? r * k + i : (r / G) * (k * G) + (r % G) + (u64)i * G;
            if (t >= tiles) break;
            const uint4 *p = in + t * 4096 + threadIdx.x;
            const uint4 a = p[0], b = p[1024], c = p[2048], d = p[3072];
            acc.x += a.x ^ b.y; acc.y += c.z ^ d.w;
        }
    }
    if (acc.x == 0x12345678u) *sink = acc;
}

// tile-style copy: 1024-thread workgroups move 64 KiB (or 128 KiB) tiles, all loads of a tile in flight,
// the next tile's loads issued before the stores of the current one (what K6 does around its sort)
template <int V>
__global__ __launch_bounds__(1024) void tile_copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 tiles)
{
    uint4 cur[V], nxt[V];
    u64 t = blockIdx.x;
    if (t >= tiles) return;
#pragma unroll
    for (int j = 0; j < V; ++j) cur[j] = in[t * (1024 * V) + j * 1024 + threadIdx.x];
    for (;;) {
        const u64 tn = t + gridDim.x;
        if (tn < tiles) {
#pragma unroll
            for (int j = 0; j < V; ++j) nxt[j] = in[tn * (1024 * V) + j * 1024 + threadIdx.x];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) out[t * (1024 * V) + j * 1024 + threadIdx.x] = cur[j];
        if (tn >= tiles) break;
#pragma unroll
        for (int j = 0; j < V; ++j) cur[j] = nxt[j];
        t = tn;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const u64 bytes = 8ull << 30, n = bytes / 16;
    uint4 *in, *out, *sink;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *name[3] = {"read ", "write", "copy "};
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {512, 1024, 2048, 4096, 8192}) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a, 0));
                if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(256), 0, 0, in, out, n, sink);
                if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(256), 0, 0, in, out, n, sink);
                if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(256), 0, 0, in, out, n, sink);
                CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            const double moved = mode == 2 ? 2.0 * bytes : (double)bytes;
            printf("%s grid %5d: %.3f ms  %.0f GB/s\n", name[mode], grid, best, moved / best / 1e6);
            fflush(stdout);
        }
    for (int grid : {256, 512})
        for (int v : {4, 8}) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a, 0));
                if (v == 4) hipLaunchKernelGGL(tile_copy_kernel<4>, dim3(grid), dim3(1024), 0, 0, in, out, n / 4096);
                else hipLaunchKernelGGL(tile_copy_kernel<8>, dim3(grid), dim3(1024), 0, 0, in, out, n / 8192);
                CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            printf("tile copy grid %4d tile %3d KiB: %.3f ms  %.0f GB/s (r+w)\n", grid, v * 16, best, 2.0 * bytes / best / 1e6);
            fflush(stdout);
        }
    for (int contig = 1; contig >= 0; --contig)
        for (int grid : {256, 512})
            for (uint32_t k : {1u, 4u, 15u, 64u}) {
                float best = 1e9;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipEventRecord(a, 0));
                    hipLaunchKernelGGL(range_read_kernel, dim3(grid), dim3(1024), 0, 0, in, bytes / 65536, k, contig, sink);
                    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (rep && ms < best) best = ms;
                }
                printf("range read %s grid %4d k %2u: %.3f ms  %.0f GB/s\n", contig ? "contiguous " : "interleaved", grid, k, best, bytes / best / 1e6);
                fflush(stdout);
            }
    return 0;
}
