// audit_kernels.hip — option "audit": every stage of a join leaves a checksum of what it read and wrote.
//
// The reference's workers meet at barriers between their phases (phj.cpp:1715-1770, cpra2.cpp:1834-1840): a tuple
// cannot be lost between two phases.  Here the phases are kernels on several streams; when a step of the multi-GPU
// slice pipeline comes out wrong (tools/stress_cpra.py: a few steps in 10^4) the final aggregates do not say WHICH
// kernel's output lacked the tuples.  With the option set, every partitioning pass and every join of a context is
// followed, on the same stream, by a read-only kernel that checks the stage's output where it lies:
//   record[stage] = { tuples that lie in a partition their key does not hash to, sum of keys, sum of payloads, tuples }
// (a stage that lost or duplicated tuples changes the sums; a stage that misplaced them raises the first word).
// The records of the last AUDIT_RING calls stay on the device (hjgpu_audit_read).  Diagnostics only: the kernels add
// one read of every relation per stage; nothing here is on a measured path, and no product path depends on it.
#include "hj_device.hpp"
#include "hj_internal.hpp"

namespace {

constexpr int AUDIT_BLOCK = 1024;

__device__ __forceinline__ void audit_flush(u64 bad, u64 sk, u64 sv, u64 seen, u64 *__restrict__ rec)
{
    bad = wave_reduce_sum(bad); sk = wave_reduce_sum(sk); sv = wave_reduce_sum(sv); seen = wave_reduce_sum(seen);
    if (hj_lane() == 0) {
        if (bad) atomicAdd(&rec[0], bad);
        atomicAdd(&rec[1], sk); atomicAdd(&rec[2], sv); atomicAdd(&rec[3], seen);
    }
}

// partition q = rows [beg[q], end ? end[q] : beg[q + 1]) of `tuples`; q's tuples must hash to q % modulo
__global__ __launch_bounds__(AUDIT_BLOCK) void audit_partitions_kernel(const u64 *__restrict__ tuples, const u64 *__restrict__ beg,
                                                                       const u64 *__restrict__ end, uint32_t parts, HjAuditHash h,
                                                                       u64 *__restrict__ rec)
{
    u64 bad = 0, sk = 0, sv = 0, seen = 0;
    for (uint32_t q = blockIdx.x; q < parts; q += gridDim.x) {
        const u64 b = beg[q], e = end ? end[q] : beg[q + 1];
        const uint32_t want = q % h.modulo;
        for (u64 j = b + threadIdx.x; j < e; j += AUDIT_BLOCK) {
            const u64 t = tuples[j];
            const uint32_t key = (uint32_t)t;
            const uint32_t p = (hj_hash(key, h.f1, h.F1) - h.p1_base) * h.F2 + hj_hash(key, h.f2, h.F2);
            if (p != want) ++bad;
            sk += key; sv += t >> 32; ++seen;
        }
    }
    audit_flush(bad, sk, sv, seen, rec);
}

__global__ __launch_bounds__(AUDIT_BLOCK) void audit_sums_packed_kernel(const u64 *__restrict__ tuples, u64 b, u64 e, u64 *__restrict__ rec)
{
    u64 sk = 0, sv = 0, seen = 0;
    for (u64 j = b + (u64)blockIdx.x * AUDIT_BLOCK + threadIdx.x; j < e; j += (u64)gridDim.x * AUDIT_BLOCK) {
        const u64 t = tuples[j];
        sk += (uint32_t)t; sv += t >> 32; ++seen;
    }
    audit_flush(0, sk, sv, seen, rec);
}

__global__ __launch_bounds__(AUDIT_BLOCK) void audit_sums_columns_kernel(const uint32_t *__restrict__ k, const uint32_t *__restrict__ v, u64 n,
                                                                         u64 *__restrict__ rec)
{
    u64 sk = 0, sv = 0, seen = 0;
    for (u64 j = (u64)blockIdx.x * AUDIT_BLOCK + threadIdx.x; j < n; j += (u64)gridDim.x * AUDIT_BLOCK) { sk += k[j]; sv += v[j]; ++seen; }
    audit_flush(0, sk, sv, seen, rec);
}

// the layout hjgpu_partition_packed_own_last_async leaves: `prefix` = plain prefix of the F counts; partitions
// [own_first, own_first + own_count) lie at the end of the n rows, the others close up (exchange_layout.hpp)
__global__ void audit_own_last_kernel(const u64 *__restrict__ prefix, uint32_t F, uint32_t own_first, uint32_t own_count, u64 n,
                                      u64 *__restrict__ beg, u64 *__restrict__ end)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= F) return;
    const u64 own_rows = prefix[own_first + own_count] - prefix[own_first];
    u64 b = prefix[p];
    if (p >= own_first + own_count) b -= own_rows;
    else if (p >= own_first) b = n - own_rows + (prefix[p] - prefix[own_first]);
    hj_store(&beg[p], b); hj_store(&end[p], b + (prefix[p + 1] - prefix[p]));
}

__global__ void audit_copy_kernel(const u64 *__restrict__ src, u64 *__restrict__ dst, uint32_t words)
{
    if (threadIdx.x < words) hj_store(&dst[threadIdx.x], src[threadIdx.x]);
}

__global__ void audit_meta_kernel(u64 *__restrict__ dst, u64 a, u64 b, u64 c, u64 d)
{
    hj_store(&dst[0], a); hj_store(&dst[1], b); hj_store(&dst[2], c); hj_store(&dst[3], d);
}

inline int launched() { return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP; }

}  // namespace

int hj_audit_partitions(const u64 *tuples, const u64 *beg, const u64 *end, uint32_t parts, const HjAuditHash &h, u64 *rec, int cus,
                        hipStream_t stream)
{
    if (!parts || !h.modulo) return HJGPU_EINVAL;
    const uint32_t grid = parts < (uint32_t)cus * 2 ? parts : (uint32_t)cus * 2;
    hipLaunchKernelGGL(audit_partitions_kernel, dim3(grid), dim3(AUDIT_BLOCK), 0, stream, tuples, beg, end, parts, h, rec);
    return launched();
}

int hj_audit_sums_packed(const u64 *tuples, u64 b, u64 e, u64 *rec, int cus, hipStream_t stream)
{
    if (e > b) hipLaunchKernelGGL(audit_sums_packed_kernel, dim3(cus * 2), dim3(AUDIT_BLOCK), 0, stream, tuples, b, e, rec);
    return launched();
}

int hj_audit_sums_columns(const uint32_t *k, const uint32_t *v, u64 n, u64 *rec, int cus, hipStream_t stream)
{
    if (n) hipLaunchKernelGGL(audit_sums_columns_kernel, dim3(cus * 2), dim3(AUDIT_BLOCK), 0, stream, k, v, n, rec);
    return launched();
}

int hj_audit_own_last(const u64 *prefix, uint32_t F, uint32_t own_first, uint32_t own_count, u64 n, u64 *beg, u64 *end, hipStream_t stream)
{
    hipLaunchKernelGGL(audit_own_last_kernel, dim3((F + 255) / 256), dim3(256), 0, stream, prefix, F, own_first, own_count, n, beg, end);
    return launched();
}

int hj_audit_copy(const u64 *src, u64 *dst, uint32_t words, hipStream_t stream)
{
    if (words > 64) return HJGPU_EINVAL;
    hipLaunchKernelGGL(audit_copy_kernel, dim3(1), dim3(64), 0, stream, src, dst, words);
    return launched();
}

int hj_audit_meta(u64 *dst, u64 a, u64 b, u64 c, u64 d, hipStream_t stream)
{
    hipLaunchKernelGGL(audit_meta_kernel, dim3(1), dim3(1), 0, stream, dst, a, b, c, d);
    return launched();
}
