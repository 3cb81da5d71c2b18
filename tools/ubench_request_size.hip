// Random reads out of a 2 GiB buffer as a function of the REQUEST size: groups of L lanes read the L 16-byte pieces of one
// pseudo-random, naturally aligned block of 16 L bytes (L = 1, 2, 4, 8, 16: 16 ... 256 bytes) with one load instruction,
// four blocks in flight per group (the NPJ probe's access shape, tools/ubench_line_sweep.py, with the block size varied).
// Question: is the ceiling of the NPJ probe a number of REQUESTS per second (then smaller buckets would not help and
// larger ones would be free) or a number of bytes / DRAM activations?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench_request_size.hip -o /tmp/urs && /tmp/urs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int L>
__global__ __launch_bounds__(256) void read_blocks(const uint4 *__restrict__ in, uint64_t blocks, uint64_t reads, uint4 *sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint32_t sub = threadIdx.x % L;
    constexpr int G = 256 / L;                                        // groups per workgroup
    const uint64_t groups = (uint64_t)gridDim.x * G, group = (uint64_t)blockIdx.x * G + threadIdx.x / L;
    for (uint64_t r = group * 4; r < reads; r += groups * 4) {
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t x = (uint64_t)(uint32_t)((uint32_t)(r + i) * 0x9E3779B1u) ^ ((r + i) >> 32);
            const uint64_t b = (uint64_t)(((unsigned __int128)(x & 0xFFFFFFFFull) * blocks) >> 32);
            v[i] = in[b * L + sub];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) *sink = acc;
}

template <int L>
static void run(const uint4 *buf, size_t bytes, uint64_t reads, uint4 *sink)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(read_blocks<L>, dim3(256 * 8), dim3(256), 0, 0, buf, (uint64_t)(bytes / (16 * L)), reads, sink);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (it && ms < best) best = ms;
    }
    printf("request %4d B: %8.3f ms for %llu reads = %6.1f G requests/s = %5.2f TB/s requested\n", 16 * L, best,
           (unsigned long long)reads, reads / best / 1e6, reads * 16.0 * L / best / 1e9);
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)2 << 30;
    const uint64_t reads = argc > 1 ? strtoull(argv[1], nullptr, 10) : 500000000ull;
    uint4 *buf, *sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    run<1>(buf, bytes, reads, sink);
    run<2>(buf, bytes, reads, sink);
    run<4>(buf, bytes, reads, sink);
    run<8>(buf, bytes, reads, sink);
    run<16>(buf, bytes, reads, sink);
    return 0;
}
