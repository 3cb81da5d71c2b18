// ubench_placement_counters.hip - what IS a "fast" allocation?  (round 4, review item 5)
// K6 pass 1 and the materialising join write 10-15 % faster into some hipMalloc blocks than into others; a streaming fill
// shows the same two kinds (1.39-1.50 vs 1.73-1.84 ms per 8.5 GB, nothing in between: profiles/r03_placement_kinds.txt).
// This program holds N fresh 8.5 GB blocks side by side, times a fill of each, and then fills the FASTEST and the
// SLOWEST of them K times through two kernels that differ in name only (fill_fast_kernel / fill_slow_kernel), so that
// a counter collection (rocprofv3 --pmc ..., tools/placement_counters.sh) attributes its counters to the two kinds.
// Last, every 256 MiB region of both blocks is filled on its own and timed: is a slow block uniformly slower, or are some
// of its regions much slower?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_placement_counters.hip -o hash_join_codes_knl_amd/lib/ubench_placement_counters
//   usage: ubench_placement_counters [blocks = 8] [launches per kind = 5]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>   // 0: probe, 1: "fast", 2: "slow" - the same code under three names
__global__ __launch_bounds__(1024) void fill_kernel(uint4 *__restrict__ out, u64 n)
{
    const uint4 v = make_uint4(KIND, 0, 0, 0);
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n; i += (u64)gridDim.x * 1024) out[i] = v;
}
// named instances (rocprofv3 reports the mangled template name: KIND is visible in it)
__global__ __launch_bounds__(1024) void fill_fast_kernel(uint4 *__restrict__ out, u64 n)
{
    const uint4 v = make_uint4(1, 0, 0, 0);
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n; i += (u64)gridDim.x * 1024) out[i] = v;
}
__global__ __launch_bounds__(1024) void fill_slow_kernel(uint4 *__restrict__ out, u64 n)
{
    const uint4 v = make_uint4(2, 0, 0, 0);
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n; i += (u64)gridDim.x * 1024) out[i] = v;
}

constexpr u64 STRIPE = 2ull << 20;

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n = argc > 1 ? atoi(argv[1]) : 8, K = argc > 2 ? atoi(argv[2]) : 5;
    const bool stripes_too = argc > 3 ? atoi(argv[3]) != 0 : true;
    const u64 bytes = 8512ull * 1000 * 1000 / STRIPE * STRIPE, stripes = bytes / STRIPE;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint4 *> blk;
    std::vector<float> ms;
    for (int i = 0; i < n; ++i) {
        uint4 *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(fill_kernel<0>, dim3(1024), dim3(1024), 0, 0, p, bytes / 16);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (rep) best = std::min(best, t);
        }
        blk.push_back(p); ms.push_back(best);
        printf("block %d at %p: fill %.3f ms = %.2f TB/s\n", i, (void *)p, best, bytes / best / 1e9);
    }
    int fast = 0, slow = 0;
    for (size_t i = 0; i < ms.size(); ++i) { if (ms[i] < ms[fast]) fast = (int)i; if (ms[i] > ms[slow]) slow = (int)i; }
    printf("fastest: block %d (%.3f ms), slowest: block %d (%.3f ms), ratio %.3f\n", fast, ms[fast], slow, ms[slow], ms[slow] / ms[fast]);
    for (int k = 0; k < K; ++k) {
        hipLaunchKernelGGL(fill_fast_kernel, dim3(1024), dim3(1024), 0, 0, blk[fast], bytes / 16);
        hipLaunchKernelGGL(fill_slow_kernel, dim3(1024), dim3(1024), 0, 0, blk[slow], bytes / 16);
    }
    CK(hipDeviceSynchronize());
    if (!stripes_too) return 0;
    // is a slow block uniformly slower?  every 256 MiB region of both blocks filled on its own, timed with events (min of 5)
    const u64 region = 256ull << 20;
    for (int which = 0; which < 2; ++which) {
        const int b = which ? slow : fast;
        std::vector<double> us;
        for (u64 off = 0; off + region <= bytes; off += region) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(fill_kernel<0>, dim3(1024), dim3(1024), 0, 0, blk[b] + off / 16, region / 16);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (rep) best = std::min(best, t);
            }
            us.push_back(best * 1e3);
        }
        std::vector<double> s = us;
        std::sort(s.begin(), s.end());
        double sum = 0;
        for (double x : s) sum += x;
        printf("%s block %d, us per 256 MiB region filled alone: mean %.1f, min %.1f, median %.1f, max %.1f (%.2f TB/s mean); in address order:",
               which ? "SLOW" : "FAST", b, sum / s.size(), s.front(), s[s.size() / 2], s.back(), region / (sum / s.size()) / 1e6);
        for (double x : us) printf(" %.0f", x);
        printf("\n");
    }
    return 0;
}
