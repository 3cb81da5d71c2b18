// How much does the NUMBER of concurrent write frontiers cost?  The materialising join writes its rows through per-wave
// cursors: 4096 resident waves x 3 result columns = 12 288 places that each receive 256 bytes at a time.  K6's passes
// write through 96-192 frontiers and reach 5.6 TB/s of read + write; the materialising join reaches 5.2.
// Every resident wave writes `iters` pieces of 256 bytes into each of 3 columns (no reads):
//   mode 0  per-wave regions: piece i of wave w of workgroup g at region(g, w) + i * 256           (12 288 frontiers)
//   mode 1  per-workgroup regions, the 8 waves' pieces interleaved: region(g) + (i * 8 + w) * 256   ( 1 536 frontiers)
//   mode 2  one region per column for everybody, pieces dealt round-robin: (i * waves + wave) * 256  (     3 frontiers)
// build + run: hipcc --offload-arch=gfx950 -O3 tools/ubench_frontiers.hip -o /tmp/ubf && /tmp/ubf
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void write_rows(uint32_t *c0, uint32_t *c1, uint32_t *c2, uint64_t col_words, uint32_t iters)
{
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = blockIdx.x;
    const uint64_t waves = (uint64_t)gridDim.x * 8, wave = (uint64_t)g * 8 + w;
    uint32_t *col[3] = {c0, c1, c2};
    for (uint32_t i = 0; i < iters; ++i) {
        uint64_t at;                                   // first word of this wave's 64-word (256-byte) piece
        if (MODE == 0) at = (wave * iters + i) * 64;
        else if (MODE == 1) at = ((uint64_t)g * iters * 8 + (uint64_t)i * 8 + w) * 64;
        else at = ((uint64_t)i * waves + wave) * 64;
#pragma unroll
        for (int c = 0; c < 3; ++c) col[c][at + lane] = (uint32_t)(at + lane + c);
    }
    (void)col_words;
}

template <int MODE>
static void run(uint32_t *c0, uint32_t *c1, uint32_t *c2, uint64_t words, uint32_t iters, const char *what)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(write_rows<MODE>, dim3(512), dim3(512), 0, 0, c0, c1, c2, words, iters);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (it && ms < best) best = ms;
    }
    const double bytes = 3.0 * 512 * 8 * (double)iters * 256;
    printf("%-62s %7.3f ms for %.1f GB = %5.2f TB/s written\n", what, best, bytes / 1e9, bytes / best / 1e9);
}

int main()
{
    const uint32_t iters = 3815;                          // 4096 waves x 3815 pieces x 64 rows = 1.0 G rows
    const uint64_t words = (uint64_t)512 * 8 * iters * 64;
    uint32_t *c[3];
    for (int i = 0; i < 3; ++i) { CHECK(hipMalloc(&c[i], words * 4)); CHECK(hipMemset(c[i], 0, words * 4)); }
    run<0>(c[0], c[1], c[2], words, iters, "per-wave regions (12 288 frontiers)");
    run<1>(c[0], c[1], c[2], words, iters, "per-workgroup regions, waves interleaved (1 536 frontiers)");
    run<2>(c[0], c[1], c[2], words, iters, "one region per column, pieces round-robin (3 frontiers)");
    return 0;
}
