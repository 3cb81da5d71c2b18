// partition_kernels.hip — K4 histogram, K5 prefix/plan, K6 scatter for gfx950.
//
// Replaces histogram()/interleave()/partition() (phj.cpp:693-772, 1263-1291,
// 1029-1231; cpra2.cpp:801-1075).  Design (not a translation):
//   * K4 reads the key column once with 16-byte loads and counts BOTH pass
//     levels at once (bin = p1*F2 + p2) in a per-workgroup LDS histogram
//     (ds_add_u32), flushed with one global atomic per non-empty bin.  The
//     reference re-reads the keys before every pass (phj.cpp:1819-1863).
//   * K5 turns the counts into absolute 64-bit offsets, atomic write cursors
//     and the tile/work-item prefix tables the later kernels walk, all on the
//     device (no host round trip between kernels).
//   * K6 counting-sorts one 8192-tuple tile by partition inside LDS (the
//     reference's per-partition write-combining buffers, phj.cpp:1115-1160,
//     become one LDS tile), claims one contiguous output run per partition
//     with a single returning atomic on the partition's cursor, and streams
//     the tile out so that consecutive lanes write consecutive addresses.
//
// Unaligned segment starts are handled by reading whole aligned 16-byte
// vectors and masking the lanes outside the segment: an aligned vector that
// overlaps a valid element never leaves that element's page.
#include "hj_device.hpp"
#include "hj_internal.hpp"

// --------------------------------------------------------------------------
// K4: fused two-level histogram.  grid = (blocks, chunks)
// --------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void hist2_kernel(
    const uint32_t *__restrict__ keys, u64 seg_beg0, u64 seg_beg1, u64 seg_beg2, u64 seg_beg3,
    u64 seg_beg4, u64 seg_beg5, u64 seg_beg6, u64 seg_beg7, u64 seg_beg8,
    uint32_t f1, uint32_t F1, uint32_t f2, uint32_t F2, u64 *__restrict__ counts)
{
    extern __shared__ uint32_t lds_hist[];
    const uint32_t P = F1 * F2;
    const uint32_t chunk = blockIdx.y;
    // chunk boundaries arrive by value (<= 8 chunks + end) to avoid a dependent load
    const u64 bounds[9] = {seg_beg0, seg_beg1, seg_beg2, seg_beg3, seg_beg4,
                           seg_beg5, seg_beg6, seg_beg7, seg_beg8};
    u64 beg = 0, end = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c == (int)chunk) { beg = bounds[c]; end = bounds[c + 1]; }

    for (uint32_t i = threadIdx.x; i < P; i += BLOCK) lds_hist[i] = 0;
    __syncthreads();

    const uint32_t a0 = (uint32_t)(((uintptr_t)keys >> 2) & 3);
    const uint4 *__restrict__ k4 = reinterpret_cast<const uint4 *>(keys - a0);
    const u64 gb = a0 + beg, ge = a0 + end;          // element coordinates from the aligned base
    const u64 v_beg = gb >> 2, v_end = (ge + 3) >> 2; // vectors [v_beg, v_end)
    const u64 stride = (u64)gridDim.x * BLOCK;
    for (u64 v = v_beg + (u64)blockIdx.x * BLOCK + threadIdx.x; v < v_end; v += stride) {
        const uint4 k = k4[v];
        const u64 g = v << 2;
        const uint32_t kk[4] = {k.x, k.y, k.z, k.w};
        if (g >= gb && g + 4 <= ge) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                atomicAdd(&lds_hist[hj_part2(kk[c], f1, F1, f2, F2)], 1u);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (g + c >= gb && g + c < ge)
                    atomicAdd(&lds_hist[hj_part2(kk[c], f1, F1, f2, F2)], 1u);
        }
    }
    __syncthreads();
    u64 *__restrict__ out = counts + (u64)chunk * P;
    for (uint32_t i = threadIdx.x; i < P; i += BLOCK) {
        const uint32_t c = lds_hist[i];
        if (c) atomicAdd(&out[i], (u64)c);
    }
}

int hj_launch_hist2(const uint32_t *keys, const u64 *seg1_host, uint32_t chunks,
                    uint32_t f1, uint32_t F1, uint32_t f2, uint32_t F2,
                    u64 *counts, int cus, hipStream_t stream)
{
    constexpr int BLOCK = 1024;
    const uint32_t P = F1 * F2;
    const size_t lds = (size_t)P * sizeof(uint32_t);
    if (chunks == 0 || chunks > 8 || lds > 128 * 1024) return HJGPU_EINVAL;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&hist2_kernel<BLOCK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return HJGPU_EHIP;
        attr_set = true;
    }
    u64 b[9];
    for (uint32_t c = 0; c <= 8; ++c) b[c] = seg1_host[c < chunks ? c : chunks];
    u64 longest = 0;
    for (uint32_t c = 0; c < chunks; ++c) if (b[c + 1] - b[c] > longest) longest = b[c + 1] - b[c];
    // enough workgroups to fill the chip (1024-thread workgroups: two per CU while
    // the histogram fits twice in LDS), never more than there are vectors
    u64 want = (longest / 4 + BLOCK - 1) / BLOCK;
    u64 cap = (u64)cus * (lds > 72 * 1024 ? 1 : 2);
    cap = (cap + chunks - 1) / chunks;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    dim3 grid((uint32_t)want, chunks);
    hipLaunchKernelGGL(hist2_kernel<BLOCK>, grid, dim3(BLOCK), lds, stream, keys,
                       b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], f1, F1, f2, F2, counts);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// --------------------------------------------------------------------------
// K5: plan.  One workgroup per job: block 0 = R, block 1 = S, block 2 = join
// work items.  A single workgroup scans up to 8*32768 counters; that is a few
// tens of microseconds and keeps every later kernel free of host round trips.
// --------------------------------------------------------------------------
constexpr int PLAN_BLOCK = 1024;

// Exclusive scan of f(i), i in [0, n), written to out[0..n] (out[n] = total).
template <typename F>
__device__ void plan_scan(uint32_t n, F f, u64 *__restrict__ out, u64 base, u64 *scratch)
{
    const uint32_t per = (n + PLAN_BLOCK - 1) / PLAN_BLOCK;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    u64 sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += f(i);
    u64 run = base + block_exclusive_scan<PLAN_BLOCK, u64>(sum, scratch);
    for (uint32_t i = lo; i < hi; ++i) { out[i] = run; run += f(i); }
    if (threadIdx.x == PLAN_BLOCK - 1) out[n] = run;   // last thread's run == base + total
    __syncthreads();
}

__device__ __forceinline__ u64 tiles_of(u64 b, u64 e, uint32_t align, uint32_t tile)
{
    if (e <= b) return 0;
    const u64 gb = (align + b) & ~3ull, ge = align + e;
    return (ge - gb + tile - 1) / tile;
}

__global__ __launch_bounds__(PLAN_BLOCK) void plan_kernel(PlanArgs a)
{
    __shared__ u64 scratch[PLAN_BLOCK / 64 + 1];
    const uint32_t P = a.F1 * a.F2;
    const uint32_t C = a.chunks;
    if (blockIdx.x < 2) {
        const int r = blockIdx.x;
        const u64 *__restrict__ cnt = a.counts[r];
        // final offsets: chunk-major flat scan == absolute positions, because the
        // chunks are contiguous, ordered, and each chunk's counts sum to its size
        plan_scan(C * P, [&](uint32_t i) { return cnt[i]; }, a.off2[r], 0, scratch);
        __threadfence_block();
        const u64 *off2 = a.off2[r];          // written above by this workgroup: no __restrict__
        for (uint32_t i = threadIdx.x; i < C * P; i += PLAN_BLOCK) a.cur2[r][i] = off2[i];
        for (uint32_t i = threadIdx.x; i <= C * a.F1; i += PLAN_BLOCK) {
            const uint32_t c = i / a.F1, p1 = i - c * a.F1;
            const u64 o = (i == C * a.F1) ? off2[C * P] : off2[(u64)c * P + (u64)p1 * a.F2];
            a.off1[r][i] = o;
            if (i < C * a.F1) a.cur1[r][i] = o;
        }
        for (uint32_t i = threadIdx.x; i <= C; i += PLAN_BLOCK)
            a.seg1[r][i] = (i == C) ? off2[C * P] : off2[(u64)i * P];
        __syncthreads();
        // pass-1 tiles: segments are the chunks of the caller's (possibly unaligned) input
        const uint32_t al = a.in_align[r];
        const uint32_t tile = a.tile;
        const u64 *seg1 = a.seg1[r];
        plan_scan(C, [&](uint32_t i) { return tiles_of(seg1[i], seg1[i + 1], al, tile); },
                  a.tp1[r], 0, scratch);
        // pass-2 tiles: segments are the pass-1 partitions inside the (aligned) workspace
        const u64 *off1 = a.off1[r];
        plan_scan(C * a.F1, [&](uint32_t i) { return tiles_of(off1[i], off1[i + 1], 0, tile); },
                  a.tp2[r], 0, scratch);
    } else {
        // join work items: partition q gets ceil(|S_q| / slice) items when both sides are non-empty
        const u64 *__restrict__ cr = a.counts[0];
        const u64 *__restrict__ cs = a.counts[1];
        const uint32_t slice = a.slice;
        auto items = [&](uint32_t q) -> u64 {
            u64 nr = 0, ns = 0;
            for (uint32_t c = 0; c < C; ++c) { nr += cr[(u64)c * P + q]; ns += cs[(u64)c * P + q]; }
            return (nr && ns) ? (ns + slice - 1) / slice : 0;
        };
        plan_scan(P, items, a.slice_prefix, 0, scratch);
        for (uint32_t q = threadIdx.x; q < P; q += PLAN_BLOCK) a.slices[q] = items(q);
    }
}

int hj_launch_plan(const PlanArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(plan_kernel, dim3(3), dim3(PLAN_BLOCK), 0, stream, a);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// Plain exclusive scan of n (<= 2^20) uint64 counters: hjgpu_partition's offsets.
__global__ __launch_bounds__(PLAN_BLOCK) void exscan_kernel(const u64 *__restrict__ in,
                                                            u64 *__restrict__ out, uint32_t n)
{
    __shared__ u64 scratch[PLAN_BLOCK / 64 + 1];
    plan_scan(n, [&](uint32_t i) { return in[i]; }, out, 0, scratch);
}

int hj_launch_exscan(const u64 *in, u64 *out, uint32_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, in, out, n);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// counts[p] = off[p+1] - off[p]: turns caller-provided partition offsets
// (hjgpu_join_partitions) back into the histogram form the plan kernel reads.
__global__ void offsets_to_counts_kernel(const u64 *__restrict__ off, u64 *__restrict__ counts, uint32_t P)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) counts[i] = off[i + 1] - off[i];
}

int hj_launch_offsets_to_counts(const u64 *off, u64 *counts, uint32_t P, hipStream_t stream)
{
    hipLaunchKernelGGL(offsets_to_counts_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, off, counts, P);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// --------------------------------------------------------------------------
// K6: scatter one pass.  Persistent workgroups walk the tile list.
// --------------------------------------------------------------------------
template <int BLOCK, int VPT>
__global__ __launch_bounds__(BLOCK) void scatter_kernel(ScatterArgs a)
{
    constexpr int TILE = BLOCK * VPT * 4;
    constexpr int NW = BLOCK / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t F = a.F;
    const uint32_t Fpad = (F + 3) & ~3u;
    u64 *delta = reinterpret_cast<u64 *>(smem);                     // [Fpad]  output - local position
    uint32_t *hist = reinterpret_cast<uint32_t *>(delta + Fpad);    // [Fpad]  counts, then local bases
    uint32_t *skeys = hist + Fpad;                                  // [TILE]
    uint32_t *svals = skeys + TILE;                                 // [TILE]
    uint32_t *wsum = svals + TILE;                                  // [NW + 1]

    const int tid = threadIdx.x;
    const u64 total_tiles = a.tile_prefix[a.nseg];
    const uint4 *__restrict__ k4 = reinterpret_cast<const uint4 *>(a.kin - a.in_align);
    const uint4 *__restrict__ v4 = reinterpret_cast<const uint4 *>(a.vin - a.in_align);
    const uint32_t factor = a.factor;
    const uint32_t bpt = (F + BLOCK - 1) / BLOCK;                   // bins per thread in the scan (<= 2)

    for (u64 t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        const uint32_t seg = hj_find_segment(a.tile_prefix, a.nseg, t);
        const u64 gb = a.in_align + a.seg_off[seg];
        const u64 ge = a.in_align + a.seg_off[seg + 1];
        const u64 g0 = (gb & ~3ull) + (t - a.tile_prefix[seg]) * (u64)TILE;

        for (uint32_t i = tid; i < F; i += BLOCK) hist[i] = 0;
        __syncthreads();

        // ---- load the tile, rank every tuple inside its partition -----------
        uint32_t key[VPT * 4], val[VPT * 4], pr[VPT * 4];
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const u64 g = g0 + (u64)(j * BLOCK + tid) * 4;
            uint4 kk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
            const bool touch = (g < ge) && (g + 4 > gb);
            if (touch) { kk = k4[g >> 2]; vv = v4[g >> 2]; }
            key[j * 4 + 0] = kk.x; key[j * 4 + 1] = kk.y; key[j * 4 + 2] = kk.z; key[j * 4 + 3] = kk.w;
            val[j * 4 + 0] = vv.x; val[j * 4 + 1] = vv.y; val[j * 4 + 2] = vv.z; val[j * 4 + 3] = vv.w;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool valid = touch && (g + c >= gb) && (g + c < ge);
                uint32_t code = 0xFFFFFFFFu;
                if (valid) {
                    const uint32_t p = hj_hash(key[j * 4 + c], factor, F);
                    const uint32_t r = atomicAdd(&hist[p], 1u);     // ds_add_rtn_u32
                    code = (p << 16) | r;
                }
                pr[j * 4 + c] = code;
            }
        }
        __syncthreads();

        // ---- local bases + one global claim per non-empty partition ----------
        uint32_t cnt[2] = {0, 0};
        uint32_t sum = 0;
        for (uint32_t i = 0; i < bpt; ++i) {
            const uint32_t bin = tid * bpt + i;
            cnt[i] = (bin < F) ? hist[bin] : 0;
            sum += cnt[i];
        }
        uint32_t run = block_exclusive_scan<BLOCK, uint32_t>(sum, wsum);
        for (uint32_t i = 0; i < bpt; ++i) {
            const uint32_t bin = tid * bpt + i;
            if (bin < F) {
                hist[bin] = run;
                if (cnt[i]) {
                    const u64 dst = atomicAdd(&a.cursors[(u64)seg * F + bin], (u64)cnt[i]);
                    delta[bin] = dst - run;
                }
                run += cnt[i];
            }
        }
        __syncthreads();
        const uint32_t tile_count = wsum[NW];

        // ---- counting sort inside LDS ----------------------------------------
#pragma unroll
        for (int e = 0; e < VPT * 4; ++e) {
            if (pr[e] != 0xFFFFFFFFu) {
                const uint32_t pos = hist[pr[e] >> 16] + (pr[e] & 0xFFFFu);
                skeys[pos] = key[e];
                svals[pos] = val[e];
            }
        }
        __syncthreads();

        // ---- stream out: lane i writes tuple i, runs are contiguous ------------
        for (uint32_t i = tid; i < tile_count; i += BLOCK) {
            const uint32_t k = skeys[i];
            const u64 d = delta[hj_hash(k, factor, F)] + i;
            a.kout[d] = k;
            a.vout[d] = svals[i];
        }
        __syncthreads();
    }
}

int hj_launch_scatter(const ScatterArgs &a, int cus, hipStream_t stream)
{
    constexpr int BLOCK = HJ_SCATTER_BLOCK, VPT = HJ_SCATTER_VPT;
    constexpr int TILE = BLOCK * VPT * 4;
    if (a.F == 0 || a.F > 2 * BLOCK || a.F > HJGPU_MAX_FANOUT) return HJGPU_EINVAL;
    const uint32_t Fpad = (a.F + 3) & ~3u;
    const size_t lds = (size_t)Fpad * 12 + (size_t)TILE * 8 + (BLOCK / 64 + 1) * 4 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&scatter_kernel<BLOCK, VPT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
            return HJGPU_EHIP;
        attr_set = true;
    }
    // two workgroups per CU fit in LDS (2 * ~77 KiB <= 160 KiB); persistent grid
    const int grid = cus * 2;
    hipLaunchKernelGGL((scatter_kernel<BLOCK, VPT>), dim3(grid), dim3(BLOCK), lds, stream, a);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}
