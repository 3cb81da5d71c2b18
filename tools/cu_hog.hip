// cu_hog.hip — diagnostics: occupies `wgs` compute units for `us` microseconds (one 1024-thread
// workgroup with 150 KiB of LDS each, spinning on the clock), to see how a kernel behaves when it
// does not get the whole chip (e.g. next to an RCCL transfer).  tools/contention_test.py uses it.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/cu_hog.hip -o hash_join_codes_knl_amd/lib/libcuhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void hog_kernel(unsigned long long ticks, unsigned int *sink)
{
    extern __shared__ unsigned int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (lds[threadIdx.x] == 0xFFFFFFFFu) *sink = 1;
}

extern "C" int hog_launch(void *stream, int wgs, int us)
{
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void *)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return 1; attr = true; }
    static unsigned int *sink = nullptr;
    if (!sink && hipMalloc(&sink, 4) != hipSuccess) return 2;
    hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(1024), 150 * 1024, (hipStream_t)stream, (unsigned long long)us * 100ull, sink);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
