// A HIP runtime small enough to run csrc/hjgpu_multi.hip - the PRODUCT's multi-GPU orchestration, the same source file -
// on a CPU, under a recorder (tests/cpp_pipeline_ordering.cpp): streams are ordered lists of operations, events are
// happens-before edges, "device" memory is host memory, every operation executes at once when it is enqueued (one legal
// order of a correctly synchronised program) while the recorder checks that every read of a buffer is ordered after the
// buffer's last write - and every write after its earlier readers and writers - through stream order, an event edge or a
// host-side wait.  Test infrastructure only; nothing in the product includes this directory.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <vector>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
struct MockStream;
struct MockEvent;
typedef MockStream *hipStream_t;
typedef MockEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
#define hipHostMallocDefault 0u
typedef void (*hipHostFn_t)(void *);

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
#define __global__
#define __device__
#define __host__
#define __launch_bounds__(...)

hipError_t hipSetDevice(int);
hipError_t hipGetDevice(int *);
hipError_t hipGetDeviceCount(int *);
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipStreamCreateWithFlags(hipStream_t *, unsigned);
hipError_t hipStreamCreateWithPriority(hipStream_t *, unsigned, int);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamQuery(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipEventCreate(hipEvent_t *);
hipError_t hipEventCreateWithFlags(hipEvent_t *, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float *, hipEvent_t, hipEvent_t);
hipError_t hipMalloc(void **, size_t);
hipError_t hipFree(void *);
hipError_t hipHostMalloc(void **, size_t, unsigned);
hipError_t hipHostFree(void *);
hipError_t hipMemcpyAsync(void *, const void *, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemcpy(void *, const void *, size_t, hipMemcpyKind);
hipError_t hipMemsetAsync(void *, int, size_t, hipStream_t);
hipError_t hipMemset(void *, int, size_t);
hipError_t hipLaunchHostFunc(hipStream_t, hipHostFn_t, void *);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t);

// a kernel launch: the recorder is told the kernel's name and its pointer / integer arguments (it knows what the few
// kernels of hjgpu_multi.hip read and write), then the kernel runs thread by thread
void mock_note_kernel(const char *name, hipStream_t stream, const std::vector<const void *> &ptrs, const std::vector<uint64_t> &ints);
inline void mock_collect(std::vector<const void *> &, std::vector<uint64_t> &) {}
template <typename T, typename... R> inline void mock_collect(std::vector<const void *> &p, std::vector<uint64_t> &i, T *a, R... r);
template <typename T, typename... R> inline void mock_collect(std::vector<const void *> &p, std::vector<uint64_t> &i, T a, R... r);
template <typename T, typename... R>
inline void mock_collect(std::vector<const void *> &p, std::vector<uint64_t> &i, T *a, R... r) { p.push_back(a); mock_collect(p, i, r...); }
template <typename T, typename... R>
inline void mock_collect(std::vector<const void *> &p, std::vector<uint64_t> &i, T a, R... r) { i.push_back((uint64_t)a); mock_collect(p, i, r...); }
template <typename... K, typename... A>
inline void mock_launch(const char *name, void (*kernel)(K...), dim3 grid, dim3 block, hipStream_t stream, A... args)
{
    std::vector<const void *> ptrs;
    std::vector<uint64_t> ints;
    mock_collect(ptrs, ints, args...);
    mock_note_kernel(name, stream, ptrs, ints);
    gridDim = grid; blockDim = block;
    for (unsigned bx = 0; bx < grid.x; ++bx)
        for (unsigned tx = 0; tx < block.x; ++tx) {
            blockIdx = dim3(bx, 0, 0); threadIdx = dim3(tx, 0, 0);
            kernel(args...);
        }
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) mock_launch(#kernel, kernel, grid, block, stream, __VA_ARGS__)

// clang's non-temporal store (the library's store policy, csrc/hj_device.hpp): a plain store on the CPU
#ifndef __clang__
#define __builtin_nontemporal_store(value, ptr) (*(ptr) = (value))
#endif
