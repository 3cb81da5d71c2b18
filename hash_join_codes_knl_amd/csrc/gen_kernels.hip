// gen_kernels.hip — counter-based relation generator and column checksums.
//
// Statistical contract of the reference generator (generate_data_for_join,
// cpra2.cpp:1578-1696; write.cpp:1482-1646): unique non-zero build keys, probe
// keys drawn from the build keys (each build key at least once when
// outer >= inner, the rest uniform picks), payload = key * odd factor, both
// columns in pseudo-random order.  The reference builds this with a serial
// MT19937 stream, a CAS hash set and a Fisher-Yates shuffle; here every tuple
// is a pure function of (seed, position), so any shard of the probe side can
// be generated independently on its own GPU with no communication
// (SURVEY.md §8f row 1).  Bit-compatibility with the MT19937 stream is NOT a
// goal (the host-side oracle generator covers that, oracle/hj_oracle.c).
#include "hj_device.hpp"
#include "hj_internal.hpp"
#include <algorithm>
#include <math.h>

// lowbias32: a bijection on 32-bit integers with mix32(0) == 0, hence
// mix32(x) != 0 for x != 0 and distinct x give distinct keys.
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ u64 splitmix64(u64 z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

struct GenArgs {
    u64 seed;
    u64 inner, inner_begin, inner_count, outer_total, outer_begin, outer_count;
    u64 distinct;               // min(inner, outer_total)  (write.cpp:1687-1689)
    u64 mul, add;               // position permutation j' = (j*mul + add) mod outer_total
    u64 mul_r, add_r;           // same for the build side, mod inner
    uint32_t key_base;          // build key i = mix32(key_base + i), key_base + i in [1, 2^32)
    uint32_t inner_factor, outer_factor;
    double zipf;                // > 0: repeat picks of the probe side follow a Zipf law over the distinct keys
    double zipf_a, zipf_b;      // rank = (a*x + 1)^b for x uniform in (0,1)   (s != 1);  distinct^x  (s == 1)
    // selectivity (write.cpp:1685-1689): the build side draws from unique[0, d), the probe side from
    // unique[d - join_d, 2d - join_d): probe rank r is build rank d - join_d + r, a match iff r < join_d
    u64 join_d, outer_shift;    // outer_shift = d - join_d
    u64 *expect;                // device [4] or NULL: count, sum key, sum key * outer_factor, sum key * inner_factor of the matching probe tuples
    uint32_t *ik, *iv, *ok, *ov;
};

__global__ __launch_bounds__(256) void generate_kernel(GenArgs a)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.ik) {
        for (u64 j = tid; j < a.inner_count; j += stride) {
            const u64 i = a.inner_begin + j;
            const u64 ip = (i * a.mul_r + a.add_r) % a.inner;
            u64 r;
            if (ip < a.distinct) r = ip;                               // every distinct key once
            else r = __umul64hi(splitmix64(ip ^ ~a.seed), a.distinct); // then repeats (outer < inner)
            const uint32_t k = mix32(a.key_base + (uint32_t)r);
            hj_store(&a.ik[j], k);
            hj_store(&a.iv[j], k * a.inner_factor);
        }
    }
    if (a.ok) {
        u64 e_n = 0, e_k = 0, e_o = 0, e_i = 0;
        for (u64 j = tid; j < a.outer_count; j += stride) {
            const u64 pos = a.outer_begin + j;
            const u64 jp = (pos * a.mul + a.add) % a.outer_total;     // "shuffle": a bijection of positions
            u64 r;
            if (jp < a.distinct) r = jp;                               // every distinct key once
            else if (a.zipf > 0.0) {
                // continuous inverse-CDF approximation of a Zipf(s) law over ranks 1..distinct
                // (the same formula as the host generator, host/write_main.cpp; write.cpp:1477-1480
                // is the reference's unfinished attempt): rank 1 is the hottest key
                const double x = ((double)(splitmix64(jp ^ a.seed) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
                const double rank = a.zipf_b == 0.0 ? pow((double)a.distinct, x) : pow(a.zipf_a * x + 1.0, a.zipf_b);
                r = rank >= 1.0 ? (u64)rank - 1 : 0;
                if (r >= a.distinct) r = a.distinct - 1;
            } else r = __umul64hi(splitmix64(jp ^ a.seed), a.distinct);  // then uniform picks
            const uint32_t k = mix32(a.key_base + (uint32_t)(a.outer_shift + r));
            hj_store(&a.ok[j], k);
            hj_store(&a.ov[j], k * a.outer_factor);
            if (r < a.join_d) { e_n += 1; e_k += k; e_o += (uint32_t)(k * a.outer_factor); e_i += (uint32_t)(k * a.inner_factor); }
        }
        if (a.expect) {
            // the aggregates the join of this probe range with the (unique-key) build side must return
            e_n = wave_reduce_sum(e_n); e_k = wave_reduce_sum(e_k); e_o = wave_reduce_sum(e_o); e_i = wave_reduce_sum(e_i);
            if (hj_lane() == 0 && e_n) {
                atomicAdd(&a.expect[0], e_n); atomicAdd(&a.expect[1], e_k);
                atomicAdd(&a.expect[2], e_o); atomicAdd(&a.expect[3], e_i);
            }
        }
    }
}

static u64 gcd_u64(u64 x, u64 y) { while (y) { u64 t = x % y; x = y; y = t; } return x; }

int hj_launch_generate(u64 seed, size_t inner, size_t inner_begin, size_t inner_count,
                       size_t outer_total, size_t outer_begin,
                       size_t outer_count, uint32_t inner_factor, uint32_t outer_factor,
                       uint32_t *ik, uint32_t *iv, uint32_t *ok, uint32_t *ov, hipStream_t stream, double zipf,
                       double selectivity, u64 *d_expect)
{
    if (!(selectivity >= 0.0) || selectivity > 1.0) return HJGPU_EINVAL;
    if (inner == 0 || inner >= 0xFFFFFFFFull) return HJGPU_EINVAL;
    if (ik && inner_begin + inner_count > inner) return HJGPU_EINVAL;
    if (ok && (outer_total == 0 || outer_begin + outer_count > outer_total)) return HJGPU_EINVAL;
    if (outer_total >= (1ull << 35)) return HJGPU_EINVAL;      // pos*mul must stay below 2^64
    GenArgs a;
    a.seed = seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull;
    a.inner = inner; a.inner_begin = inner_begin; a.inner_count = inner_count;
    a.outer_total = outer_total ? outer_total : 1;
    a.outer_begin = outer_begin; a.outer_count = outer_count;
    u64 mul = 402653189ull;                                    // prime < 2^29
    while (gcd_u64(mul, a.outer_total) != 1) mul += 2;
    a.mul = mul;
    u64 mul_r = 268435459ull;                                  // prime > 2^28
    while (gcd_u64(mul_r, (u64)inner) != 1) mul_r += 2;
    a.mul_r = mul_r;
    a.add_r = (a.seed >> 13) % inner;
    a.distinct = (outer_total && outer_total < inner) ? outer_total : inner;
    a.add = (a.seed >> 7) % a.outer_total;
    a.join_d = (u64)((double)a.distinct * selectivity);           // write.cpp:1689
    if (a.join_d > a.distinct) a.join_d = a.distinct;
    a.outer_shift = a.distinct - a.join_d;
    a.expect = d_expect;
    if (2 * a.distinct - a.join_d >= 0xFFFFFFFFull) return HJGPU_EINVAL;
    a.key_base = 1u + (uint32_t)((a.seed >> 11) % (0xFFFFFFFFull - (2 * a.distinct - a.join_d)));
    a.inner_factor = inner_factor | 1u; a.outer_factor = outer_factor | 1u;
    a.zipf = zipf > 0.0 ? zipf : 0.0; a.zipf_a = a.zipf_b = 0.0;
    if (a.zipf > 0.0 && fabs(a.zipf - 1.0) >= 1e-9) {
        a.zipf_a = pow((double)a.distinct, 1.0 - a.zipf) - 1.0;
        a.zipf_b = 1.0 / (1.0 - a.zipf);
    }
    a.ik = ik; a.iv = iv; a.ok = ok; a.ov = ov;
    hipLaunchKernelGGL(generate_kernel, dim3(4096), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// sums[0] += key, sums[1] += key*fa, sums[2] += key*fb  (products mod 2^32)
// Also the "simple dwordx4 read kernel" of SURVEY 8d: 16-byte loads, 4 per lane in flight,
// nothing written - bench.py times it for the empirical streaming-read ceiling of the box.
__global__ __launch_bounds__(256) void column_sums_kernel(const uint32_t *__restrict__ keys, u64 n,
                                                          uint32_t fa, uint32_t fb, u64 *sums)
{
    __shared__ u64 red[3][4];
    u64 s0 = 0, s1 = 0, s2 = 0;
    auto add = [&](uint32_t k) { s0 += k; s1 += (uint32_t)(k * fa); s2 += (uint32_t)(k * fb); };
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    // scalar head up to the first 16-byte boundary, vector body, scalar tail
    const u64 head = min(n, (u64)((16 - ((uintptr_t)keys & 15)) & 15) / 4);
    const uint4 *v = (const uint4 *)(keys + head);
    const u64 nv = (n - head) / 4;
    u64 i = tid;
    for (; i + 3 * stride < nv; i += 4 * stride) {
        const uint4 a = v[i], b = v[i + stride], c = v[i + 2 * stride], d = v[i + 3 * stride];
        add(a.x); add(a.y); add(a.z); add(a.w); add(b.x); add(b.y); add(b.z); add(b.w);
        add(c.x); add(c.y); add(c.z); add(c.w); add(d.x); add(d.y); add(d.z); add(d.w);
    }
    for (; i < nv; i += stride) { const uint4 a = v[i]; add(a.x); add(a.y); add(a.z); add(a.w); }
    if (tid < head) add(keys[tid]);
    const u64 done = head + nv * 4;
    if (tid < n - done) add(keys[done + tid]);
    s0 = wave_reduce_sum(s0); s1 = wave_reduce_sum(s1); s2 = wave_reduce_sum(s2);
    const int wave = threadIdx.x >> 6;
    if (hj_lane() == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        u64 s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        atomicAdd(&sums[threadIdx.x], s);
    }
}

int hj_launch_column_sums(const uint32_t *keys, size_t n, uint32_t fa, uint32_t fb, u64 *sums3,
                          hipStream_t stream)
{
    if (hj_zero_async(sums3, 3 * sizeof(u64), stream) != hipSuccess) return HJGPU_EHIP;
    hipLaunchKernelGGL(column_sums_kernel, dim3(2048), dim3(256), 0, stream, keys, (u64)n, fa, fb, sums3);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// The "simple dwordx4 read kernel" of SURVEY 8d: 1024-thread workgroups sweep 64 KiB pieces, four
// 16-byte loads per lane in flight, the data only XORed together (column_sums_kernel's two 32-bit
// multiplies per key make IT instruction-bound at ~5.2 TB/s; this one reads at 6.6-6.7 TB/s).
__global__ __launch_bounds__(1024) void stream_read_kernel(const uint4 *__restrict__ in, u64 pieces, uint4 *sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u64 t = blockIdx.x; t < pieces; t += gridDim.x) {
        const uint4 *p = in + t * 4096 + threadIdx.x;
        const uint4 a = p[0], b = p[1024], c = p[2048], d = p[3072];
        acc.x ^= a.x ^ b.y; acc.y ^= c.z ^ d.w; acc.z ^= a.z ^ c.x; acc.w ^= b.w ^ d.y;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) hj_store(sink, acc);      // keeps the loads alive
}

int hj_launch_stream_read(const void *p, size_t bytes, void *sink16, int cus, hipStream_t stream)
{
    const u64 pieces = bytes / 65536;
    if (pieces == 0 || ((uintptr_t)p & 15)) return HJGPU_EINVAL;
    hipLaunchKernelGGL(stream_read_kernel, dim3(cus * 2), dim3(1024), 0, stream, (const uint4 *)p, pieces, (uint4 *)sink16);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// Random 64-byte line reads (hjgpu_random_line_read_ms): the access shape of the NPJ probe without the join - the four
// lanes of a quad fetch the four quarters of one pseudo-random line of the buffer with ONE load instruction, four lines
// in flight per quad.  What it reaches is the memory system's rate for independent 64-byte reads out of a table that
// does not fit the caches (requests per second, not bytes, are the limit: profiles/r03_request_size.txt), i.e. the ceiling bench.py prices NPJ against.
__global__ __launch_bounds__(256) void random_line_read_kernel(const uint4 *__restrict__ in, u64 lines, u64 reads, uint4 *sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint32_t sub = threadIdx.x & 3;
    const u64 quads = (u64)gridDim.x * 64, quad = (u64)blockIdx.x * 64 + (threadIdx.x >> 2);
    for (u64 r = quad * 4; r < reads; r += quads * 4) {
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // a multiplicative hash of the read's index, as the join's H(key, f, lines)
            const u64 x = (u64)(uint32_t)((uint32_t)(r + i) * 0x9E3779B1u) ^ ((r + i) >> 32);
            const u64 line = (u64)(((unsigned __int128)(x & 0xFFFFFFFFull) * lines) >> 32);
            v[i] = in[4 * line + sub];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) hj_store(sink, acc);      // keeps the loads alive
}

int hj_launch_random_line_read(const void *p, size_t bytes, size_t reads, void *sink16, int cus, hipStream_t stream)
{
    const u64 lines = bytes / 64;
    if (lines == 0 || reads == 0 || ((uintptr_t)p & 63)) return HJGPU_EINVAL;
    hipLaunchKernelGGL(random_line_read_kernel, dim3(cus * 8), dim3(256), 0, stream, (const uint4 *)p, lines, (u64)reads, (uint4 *)sink16);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// Random 8-byte compare-and-swaps (hjgpu_random_cas_ms): the access shape of the NPJ build (npj.cpp:196-210: one CAS of an
// empty bucket per build tuple) without the join - `ops` independent pseudo-random buckets of a zeroed buffer, each claimed
// with ONE 64-bit CAS (expected 0), U of them in flight per lane; LOAD: a plain load of the bucket first, as the build
// does to skip taken buckets.  What it reaches is the memory system's rate for independent returning atomics.
template <int U, bool LOAD>
__global__ __launch_bounds__(256) void random_cas_kernel(u64 *__restrict__ table, u64 buckets, u64 ops, u64 *sink)
{
    u64 acc = 0;
    const u64 stride = (u64)gridDim.x * 256 * U;
    for (u64 r = ((u64)blockIdx.x * 256 + threadIdx.x) * U; r < ops; r += stride) {
        u64 at[U], seen[U], old[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const u64 x = (u64)(uint32_t)((uint32_t)(r + i) * 0x9E3779B1u) ^ ((r + i) >> 32);
            at[i] = (u64)(((unsigned __int128)(x & 0xFFFFFFFFull) * buckets) >> 32);
            seen[i] = 0;
        }
        if (LOAD) {
#pragma unroll
            for (int i = 0; i < U; ++i) seen[i] = __builtin_nontemporal_load(&table[at[i]]);
        }
        // all U atomics are issued back to back (no predicate around them: an operation past `ops` or on a bucket that was
        // seen taken compares with a value no bucket holds and changes nothing)
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const bool live = r + i < ops && (uint32_t)seen[i] == 0u;
            old[i] = atomicCAS(&table[at[i]], live ? 0ull : ~0ull, live ? (((r + i) << 1) | 1ull) : ~0ull);
        }
#pragma unroll
        for (int i = 0; i < U; ++i) acc ^= old[i];
    }
    if (acc == 0x9E3779B97F4A7C15ull) *sink = acc;          // keeps the returns alive
}

int hj_launch_random_cas(void *p, size_t bytes, size_t ops, int in_flight, bool load_first, void *sink8, int cus, hipStream_t stream)
{
    const u64 buckets = bytes / 8;
    if (buckets == 0 || ops == 0 || ((uintptr_t)p & 7)) return HJGPU_EINVAL;
    u64 *t = (u64 *)p, *sk = (u64 *)sink8;
    const dim3 grid(cus * 16), block(256);
#define RC(U, L) hipLaunchKernelGGL((random_cas_kernel<U, L>), grid, block, 0, stream, t, buckets, (u64)ops, sk)
    if (in_flight == 1) { if (load_first) RC(1, true); else RC(1, false); }
    else if (in_flight == 2) { if (load_first) RC(2, true); else RC(2, false); }
    else if (in_flight == 4) { if (load_first) RC(4, true); else RC(4, false); }
    else if (in_flight == 8) { if (load_first) RC(8, true); else RC(8, false); }
    else return HJGPU_EINVAL;
#undef RC
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// Placement probe (hjgpu_api.hip, ensure_placed): a plain streaming fill of a freshly allocated buffer.  Its rate
// differs by up to 26 % between allocations of the same size on one MI355X (physical placement; profiles/r02_placement.txt)
// and predicts how fast K6 pass 1 will write into that buffer.
__global__ __launch_bounds__(1024) void fill_probe_kernel(uint4 *__restrict__ out, u64 n)
{
    const uint4 v = make_uint4(0, 0, 0, 0);
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n; i += (u64)gridDim.x * 1024) out[i] = v;
}

// Result rows -> page-locked host memory by a KERNEL (16-byte stores over PCIe; few workgroups: the bus, not the CUs, is the
// limit).  hjgpu_join_host_rows sends a batch's rows home while the next batch is uploaded: with hipMemcpyAsync in both
// directions the runtime put upload and download on the same DMA engine from the second call of a process on (one after
// the other: 430 ms instead of 255 for 8.5 GB up and 12 GB down); the DMA engines now carry the upload only.
__global__ __launch_bounds__(256) void copy_to_host_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, u64 n)
{
    // 32-bit elements (a batch's rows start wherever the batches before it ended): a wave's store is 256 contiguous bytes
    const u64 stride = (u64)gridDim.x * 256;
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) hj_store(&dst[i + j * stride], v[j]);
    }
    for (; i < n; i += stride) hj_store(&dst[i], src[i]);
}

int hj_launch_copy_to_host(void *host_mapped, const void *d, size_t bytes, hipStream_t stream)
{
    if (!bytes) return HJGPU_OK;
    if (((uintptr_t)host_mapped & 3) || ((uintptr_t)d & 3) || (bytes & 3)) return HJGPU_EALIGN;
    hipLaunchKernelGGL(copy_to_host_kernel, dim3(64), dim3(256), 0, stream, (uint32_t *)host_mapped, (const uint32_t *)d, (u64)(bytes / 4));
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// Clears and device-to-device copies of the library's own state and of the caller's columns: own kernels, because the runtime's
// (hipMemsetAsync, hipMemcpyAsync) store plainly and the library's store policy (hj_device.hpp) has no plain store beside other
// queues' work - NPJ's 2 GB table clear writes as many bytes as a K6 pass over the build side.  Any 4-byte aligned range; the
// body in 16-byte non-temporal stores.
__global__ __launch_bounds__(256) void zero_kernel(uint32_t *__restrict__ p, u64 words, uint32_t word)
{
    // words [0, head) up to the first 16-byte boundary, 16-byte vectors, then the tail
    const u64 head = min(words, (u64)((16 - ((uintptr_t)p & 15)) & 15) / 4);
    const u64 vecs = (words - head) / 4, tail0 = head + vecs * 4;
    uint4 *v = reinterpret_cast<uint4 *>(p + head);
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, stride = (u64)gridDim.x * 256;
    for (u64 i = tid; i < vecs; i += stride) hj_store(&v[i], make_uint4(word, word, word, word));
    if (tid < head) hj_store(&p[tid], word);
    if (tid < words - tail0) hj_store(&p[tail0 + tid], word);
}

__global__ __launch_bounds__(256) void copy_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, u64 words, uint32_t vec)
{
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, stride = (u64)gridDim.x * 256;
    if (vec) {              // both 16-byte aligned: 4 vectors in flight per lane
        const u64 vecs = words / 4;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        u64 i = tid;
        for (; i + 3 * stride < vecs; i += 4 * stride) {
            const uint4 a = hj_load_nt(s4 + i), b = hj_load_nt(s4 + i + stride), c = hj_load_nt(s4 + i + 2 * stride), d = hj_load_nt(s4 + i + 3 * stride);
            hj_store(&d4[i], a); hj_store(&d4[i + stride], b); hj_store(&d4[i + 2 * stride], c); hj_store(&d4[i + 3 * stride], d);
        }
        for (; i < vecs; i += stride) hj_store(&d4[i], hj_load_nt(s4 + i));
        if (tid < words - vecs * 4) hj_store(&dst[vecs * 4 + tid], src[vecs * 4 + tid]);
    } else
        for (u64 i = tid; i < words; i += stride) hj_store(&dst[i], src[i]);
}

hipError_t hj_fill_async(void *p, uint32_t word, size_t bytes, hipStream_t stream)
{
    if (!bytes) return hipSuccess;
    if (((uintptr_t)p & 3) || (bytes & 3)) return word == 0 ? hipMemsetAsync(p, 0, bytes, stream) : hipErrorInvalidValue;      // (no such caller: every clear is of 4-byte words)
    const u64 words = bytes / 4;
    const unsigned grid = (unsigned)std::min<u64>((words / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, stream, (uint32_t *)p, words, word);
    return hipGetLastError();
}

hipError_t hj_zero_async(void *p, size_t bytes, hipStream_t stream) { return hj_fill_async(p, 0u, bytes, stream); }

hipError_t hj_copy_async(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (!bytes) return hipSuccess;
    if (((uintptr_t)dst & 3) || ((uintptr_t)src & 3) || (bytes & 3)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    const u64 words = bytes / 4;
    const uint32_t vec = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0 ? 1u : 0u;
    const unsigned grid = (unsigned)std::min<u64>((words / (vec ? 16 : 1) + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, stream, (uint32_t *)dst, (const uint32_t *)src, words, vec);
    return hipGetLastError();
}

int hj_launch_fill_probe(void *p, size_t bytes, hipStream_t stream)
{
    if (((uintptr_t)p & 15) || bytes < 16) return HJGPU_EINVAL;
    hipLaunchKernelGGL(fill_probe_kernel, dim3(1024), dim3(1024), 0, stream, (uint4 *)p, (u64)(bytes / 16));
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}
