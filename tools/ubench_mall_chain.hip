// ubench_mall_chain.hip — can an intermediate relation live in the 256 MiB Infinity Cache instead of HBM?
//
// PHJ moves every tuple through an intermediate copy (pass 1 writes it, pass 2 reads it): 16 of the 44 bytes per
// probe tuple.  If the relation is processed in batches whose intermediate piece fits the memory-side cache, and the
// SAME intermediate buffer is reused by every batch, its lines might never have to reach HBM.  This measures the
// best case with plain copies:   A (8 GiB) --copy--> T (batch-sized, reused) --copy--> B (8 GiB)
// against the two-pass reference   A --copy--> T8 (8 GiB) --copy--> B,  as back-to-back launches on one stream,
// and as ONE persistent kernel that alternates the two copies per batch behind a grid barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mall_chain.hip -o hash_join_codes_knl_amd/lib/ubench_mall_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS>
__device__ __forceinline__ void copy_span(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n, u64 first, u64 step)
{
    for (u64 i = first; i < n; i += step * 4) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + j * step < n) {
                if (NTL) { const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(in + i + j * step)); v[j] = make_uint4(t.x, t.y, t.z, t.w); }
                else v[j] = in[i + j * step];
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + j * step < n) {
                if (NTS) { v4u t; t.x = v[j].x; t.y = v[j].y; t.z = v[j].z; t.w = v[j].w; __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(out + i + j * step)); }
                else out[i + j * step] = v[j];
            }
    }
}

template <bool NTL, bool NTS>
__global__ __launch_bounds__(1024) void copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n)
{
    copy_span<NTL, NTS>(in, out, n, (u64)blockIdx.x * 1024 + threadIdx.x, (u64)gridDim.x * 1024);
}

// one persistent kernel: for every batch, A -> T (input non-temporal: it is read once), grid barrier, T -> B (output
// non-temporal), grid barrier.  The barrier is a monotonic counter polled with sc1 loads.
__global__ __launch_bounds__(1024) void chain_kernel(const uint4 *__restrict__ a, uint4 *__restrict__ t, uint4 *__restrict__ b,
                                                     u64 n, u64 batch, unsigned *counter)
{
    const u64 first = (u64)blockIdx.x * 1024 + threadIdx.x, step = (u64)gridDim.x * 1024;
    unsigned target = 0;
    auto grid_barrier = [&]() {
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(counter, 1u);
            target += gridDim.x;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    };
    for (u64 at = 0; at < n; at += batch) {
        const u64 m = n - at < batch ? n - at : batch;
        copy_span<true, false>(a + at, t, m, first, step);
        grid_barrier();
        copy_span<false, true>(t, b + at, m, first, step);
        grid_barrier();
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main()
{
    const u64 bytes = 8ull << 30, n = bytes / 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint4 *a, *b, *t8;
    unsigned *counter;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&t8, bytes)); CK(hipMalloc(&counter, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(t8, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](auto fn) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0)); fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        CK(hipGetLastError());
        return best;
    };
    float ms = timed([&] {
        hipLaunchKernelGGL((copy_kernel<false, false>), dim3(cus), dim3(1024), 0, 0, a, t8, n);
        hipLaunchKernelGGL((copy_kernel<false, false>), dim3(cus), dim3(1024), 0, 0, t8, b, n);
    });
    printf("reference: A -> T8 (8 GiB) -> B, two launches                         %8.3f ms  (%.0f GB/s over 32 GiB of traffic)\n", ms, 4.0 * bytes / ms / 1e6);
    fflush(stdout);
    for (u64 mib : {16ull, 32ull, 64ull, 96ull, 128ull, 192ull, 512ull}) {
        const u64 batch = (mib << 20) / 16;
        ms = timed([&] {
            for (u64 at = 0; at < n; at += batch) {
                const u64 m = n - at < batch ? n - at : batch;
                hipLaunchKernelGGL((copy_kernel<true, false>), dim3(cus), dim3(1024), 0, 0, a + at, t8, m);
                hipLaunchKernelGGL((copy_kernel<false, true>), dim3(cus), dim3(1024), 0, 0, t8, b + at, m);
            }
        });
        printf("batches of %4llu MiB through ONE reused buffer, 2 launches per batch     %8.3f ms  (%.0f GB/s if only A and B touch HBM)\n",
               mib, ms, 2.0 * bytes / ms / 1e6);
        fflush(stdout);
        CK(hipMemset(counter, 0, 4));
        ms = timed([&] {
            CK(hipMemsetAsync(counter, 0, 4, 0));
            hipLaunchKernelGGL(chain_kernel, dim3(cus), dim3(1024), 0, 0, a, t8, b, n, batch, counter);
        });
        printf("batches of %4llu MiB through ONE reused buffer, one persistent kernel    %8.3f ms  (%.0f GB/s if only A and B touch HBM)\n",
               mib, ms, 2.0 * bytes / ms / 1e6);
        fflush(stdout);
    }
    return 0;
}
