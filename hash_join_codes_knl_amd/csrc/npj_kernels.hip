// npj_kernels.hip — K2 build, K3 probe, K9 close_gaps for the no-partition join.
//
// Replaces build()/probe()/close_gaps() of npj.cpp:190-212, 216-364 (scalar
// definition 412-445), 475-514.  One global linear-probing table of
// uint64 (payload << 32 | key), empty = 0, exactly the reference's bucket
// format, so a table built here can be probed by the oracle and vice versa.
// Design (not a translation):
//   * build: one tuple per lane, 64-bit global compare-and-swap against 0; the
//     key/payload columns are read with 16-byte loads.
//   * probe: one tuple per lane per chain, four chains per lane from a 16-byte
//     load, each walking consecutive buckets to the first empty one and
//     reporting every key match (the reference's "refill finished lanes" loop,
//     npj.cpp:251-254, exists because its 16 lanes are all it has; here
//     thousands of resident waves hide the divergence instead).
//   * The load factor is a free knob (results do not depend on it): the GPU
//     default is 0.5, the reference's 0.90 (npj.cpp:944) gives ~59-bucket walks.
#include "hj_device.hpp"
#include "hj_internal.hpp"
#include "hj_emit.hpp"

__device__ __forceinline__ u64 npj_bucket(uint32_t key, uint32_t factor, u64 buckets)
{
    // h = ((uint64)(uint32)(key*factor) * buckets) >> 32, buckets may exceed 2^32
    const u64 x = (u64)(uint32_t)(key * factor);
    return (u64)(((unsigned __int128)x * buckets) >> 32);
}

__global__ __launch_bounds__(256) void npj_build_kernel(const uint32_t *__restrict__ keys,
                                                        const uint32_t *__restrict__ vals, u64 n,
                                                        u64 *table, u64 buckets, uint32_t factor,
                                                        uint32_t *zero_key_flag, uint32_t line_hash)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t key = keys[i];
        if (key == 0) { atomicOr(zero_key_flag, 1u); continue; }   // npj.cpp:583: 0 is "empty"
        const u64 pair = ((u64)vals[i] << 32) | key;
        // line_hash: walks start on a 64-byte line of 8 buckets (see npj_probe_line_kernel)
        u64 h = line_hash ? npj_bucket(key, factor, buckets >> 3) << 3 : npj_bucket(key, factor, buckets);
        for (;;) {
            // skip the buckets that are visibly taken with plain loads (after the first one the line is
            // in L2; a stale "empty" only costs the failed CAS below, a bucket never becomes empty again):
            // every failed CAS is a round trip to the memory side
            while ((uint32_t)table[h] != 0u) { if (++h == buckets) h = 0; }
            // claim the first bucket whose low word is empty (npj.cpp:204-210)
            const u64 old = atomicCAS(&table[h], 0ull, pair);
            if (old == 0ull) break;
            if (++h == buckets) h = 0;
        }
    }
}

int hj_launch_npj_build(const uint32_t *keys, const uint32_t *vals, size_t n, u64 *table,
                        size_t buckets, uint32_t factor, uint32_t *zero_key_flag,
                        int cus, hipStream_t stream, bool line_hash)
{
    if (buckets <= n) return HJGPU_EINVAL;       // a walk must always find an empty bucket
    if (line_hash && (buckets % 8 != 0 || ((uintptr_t)table & 63))) return HJGPU_EINVAL;
    u64 blocks = (n + 255) / 256;
    if (blocks > (u64)cus * 16) blocks = (u64)cus * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(npj_build_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, keys, vals,
                       (u64)n, table, (u64)buckets, factor, zero_key_flag, line_hash ? 1u : 0u);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

constexpr int NPJ_PROBE_BLOCK = 256;

// GROUPED: the walk fetches aligned groups of 4 buckets (32 bytes, two 16-byte loads
// issued together) instead of one bucket per dependent load.  The walk has to reach
// the first EMPTY bucket (every match counts, npj.cpp:426-442), i.e. 2.5 buckets on
// average at load 0.5: with one bucket per load that is 2.5 dependent memory round
// trips per probe; a group resolves most walks in one.  Needs buckets % 4 == 0
// (the library's own tables; any other table takes the bucket-at-a-time path).
template <bool GROUPED>
__global__ __launch_bounds__(NPJ_PROBE_BLOCK) void npj_probe_kernel(NpjProbeArgs a)
{
    constexpr int NW = NPJ_PROBE_BLOCK / 64;
    __shared__ u64 red[4][NW];
    __shared__ u64 wave_cursor[NW];
    const int wave = threadIdx.x >> 6;
    Emitter em;
    em.init(a.ok, a.oov, a.oiv, a.block_size, a.block_limit, a.block_counter, a.overflow,
            &wave_cursor[wave]);
    if (hj_lane() == 0) wave_cursor[wave] = HJ_NO_CURSOR;

    const uint32_t a0 = (uint32_t)(((uintptr_t)a.keys >> 2) & 3);
    const uint4 *__restrict__ k4 = reinterpret_cast<const uint4 *>(a.keys - a0);
    const uint4 *__restrict__ v4 = reinterpret_cast<const uint4 *>(a.vals - a0);
    const u64 gb = a0, ge = a0 + a.n;
    const u64 nvec = (ge + 3) >> 2;
    const u64 stride = (u64)gridDim.x * NPJ_PROBE_BLOCK;
    const u64 *__restrict__ table = a.table;
    const u64 buckets = a.buckets;
    const uint32_t factor = a.factor;

    u64 acc_n = 0, acc_k = 0, acc_o = 0, acc_i = 0;
    for (u64 v = (u64)blockIdx.x * NPJ_PROBE_BLOCK + threadIdx.x; v < nvec; v += stride) {
        const uint4 kk = k4[v], vv = v4[v];
        const u64 g = v << 2;
        const uint32_t key[4] = {kk.x, kk.y, kk.z, kk.w};
        const uint32_t val[4] = {vv.x, vv.y, vv.z, vv.w};
        u64 h[4];
        bool act[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            act[j] = (g + j >= gb) && (g + j < ge);
            h[j] = npj_bucket(key[j], factor, buckets);
        }
        if (GROUPED) {
            const uint4 *__restrict__ t4 = reinterpret_cast<const uint4 *>(table);
            while (act[0] | act[1] | act[2] | act[3]) {
                uint4 lo[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {           // all group loads of the 4 chains in flight together
                    lo[j] = make_uint4(0, 0, 0, 0); hi[j] = lo[j];
                    if (act[j]) { const u64 grp = h[j] >> 2; lo[j] = t4[2 * grp]; hi[j] = t4[2 * grp + 1]; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (act[j]) {
                        const uint32_t bk[4] = {lo[j].x, lo[j].z, hi[j].x, hi[j].z};   // keys of the group
                        const uint32_t bv[4] = {lo[j].y, lo[j].w, hi[j].y, hi[j].w};   // payloads
                        const uint32_t first = (uint32_t)h[j] & 3u;
                        bool open = true;                                              // no empty bucket seen yet
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const bool in = open && ((uint32_t)b >= first);
                            if (in && bk[b] == 0u) open = false;
                            else if (in && bk[b] == key[j]) {
                                acc_n += 1; acc_k += key[j]; acc_o += val[j]; acc_i += bv[b];
                                em.emit(key[j], val[j], bv[b]);
                            }
                        }
                        if (!open) act[j] = false;
                        else { h[j] = (h[j] & ~3ull) + 4; if (h[j] >= buckets) h[j] = 0; }
                    }
                }
            }
        } else {
            u64 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = act[j] ? table[h[j]] : 0ull;
            while (act[0] | act[1] | act[2] | act[3]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (act[j]) {
                        if ((uint32_t)t[j] == 0u) {
                            act[j] = false;
                        } else {
                            if ((uint32_t)t[j] == key[j]) {
                                const uint32_t iv = (uint32_t)(t[j] >> 32);
                                acc_n += 1; acc_k += key[j]; acc_o += val[j]; acc_i += iv;
                                em.emit(key[j], val[j], iv);
                            }
                            if (++h[j] == buckets) h[j] = 0;
                            t[j] = table[h[j]];
                        }
                    }
                }
            }
        }
    }
    if (a.ok && hj_lane() == 0)
        a.final_offsets[(u64)blockIdx.x * NW + wave] = wave_cursor[wave];
    acc_n = wave_reduce_sum(acc_n); acc_k = wave_reduce_sum(acc_k);
    acc_o = wave_reduce_sum(acc_o); acc_i = wave_reduce_sum(acc_i);
    if (hj_lane() == 0) { red[0][wave] = acc_n; red[1][wave] = acc_k; red[2][wave] = acc_o; red[3][wave] = acc_i; }
    __syncthreads();
    if (threadIdx.x < 4) {
        u64 s = 0;
        for (int i = 0; i < NW; ++i) s += red[threadIdx.x][i];
        if (s) atomicAdd(reinterpret_cast<u64 *>(a.result) + threadIdx.x, s);
    }
}


// LINE table (the library's own whole joins, hjgpu_npj*): the walk of a key starts on the
// 64-byte line of 8 buckets it hashes to, h = 8 * H(key, f, buckets / 8), instead of on an
// arbitrary bucket.  Same bucket format, same CAS build, same walk to the first empty bucket
// with every match reported (npj.cpp:204-210, 426-442) - only the hash range differs, and like
// the load factor it does not change the join result.
// The probe is COOPERATIVE: the four lanes of a quad fetch the four 16-byte quarters of one
// key's line in ONE load instruction, so the line is one L2 request and one memory-side
// request.  PMC at 64M x 1G, load 0.25 (TCC_REQ / TCC_EA0_RDREQ per probe, probe time):
//   reference hash, 32-byte groups, 2 loads per lane and chain   1.83 / 1.48   27.4 ms
//   line hash, one lane reads its whole line with 4 loads        3.59 / 1.06   29.9 ms
// a successful probe has to see the bucket AFTER its match, so with arbitrary start buckets
// every fourth probe needs a second group; and every extra load instruction is an extra L2
// request even when it hits the same line (fit: 12.9 ms per G memory-side requests + 4.5 ms
// per G L2 requests).
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xF, 0xF, true);
}

// MATERIALIZE is a template parameter on purpose: with a run-time `if (a.ok)` inside the walk loop
// the -O3 build evaluated that (uniform) test per lane at a loop header that the continuation
// re-enters with only the walking lanes enabled, and later rounds of other lanes then took the
// emit path with a.ok == NULL (memory access fault; -O1 was fine).
template <bool MATERIALIZE>
__global__ __launch_bounds__(NPJ_PROBE_BLOCK) void npj_probe_line_kernel(NpjProbeArgs a)
{
    constexpr int NW = NPJ_PROBE_BLOCK / 64;
    constexpr int B = 4;                                   // lines in flight per quad
    __shared__ u64 red[4][NW];
    __shared__ u64 wave_cursor[NW];
    const int wave = threadIdx.x >> 6;
    Emitter em;
    em.init(a.ok, a.oov, a.oiv, a.block_size, a.block_limit, a.block_counter, a.overflow,
            &wave_cursor[wave]);
    if (hj_lane() == 0) wave_cursor[wave] = HJ_NO_CURSOR;

    const uint32_t a0 = (uint32_t)(((uintptr_t)a.keys >> 2) & 3);
    const uint4 *__restrict__ k4 = reinterpret_cast<const uint4 *>(a.keys - a0);
    const uint4 *__restrict__ v4 = reinterpret_cast<const uint4 *>(a.vals - a0);
    const u64 gb = a0, ge = a0 + a.n;
    const u64 nvec = (ge + 3) >> 2;
    const u64 stride = (u64)gridDim.x * NPJ_PROBE_BLOCK;
    const uint4 *__restrict__ t4 = reinterpret_cast<const uint4 *>(a.table);
    const u64 lines = a.buckets >> 3;
    const uint32_t factor = a.factor;
    const uint32_t sub = threadIdx.x & 3;                  // my quarter of the line: buckets 2*sub, 2*sub + 1

    u64 acc_n = 0, acc_k = 0, acc_o = 0, acc_i = 0;
    // whole waves iterate together (the quad exchanges below need all four lanes)
    for (u64 v0 = (u64)blockIdx.x * NPJ_PROBE_BLOCK + (threadIdx.x & ~63u); v0 < nvec; v0 += stride) {
        const u64 v = v0 + hj_lane();
        uint4 kk = make_uint4(0, 0, 0, 0), vv = kk;
        if (v < nvec) { kk = k4[v]; vv = v4[v]; }
        const u64 g = v << 2;
        const uint32_t kc[4] = {kk.x, kk.y, kk.z, kk.w}, vc[4] = {vv.x, vv.y, vv.z, vv.w};
        uint32_t okc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) okc[j] = (v < nvec && g + j >= gb && g + j < ge) ? 1u : 0u;

        // the quad's 16 tuples (4 lanes x 4 components), B at a time
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += B) {
            uint32_t key[B], val[B];
            bool act[B];
            u64 ln[B];
            uint4 q[B];
#pragma unroll
            for (int i = 0; i < B; ++i) {
                const int r = r0 + i, comp = r & 3;                     // static after unrolling
                // the tuple's owner is lane r / 4 of the quad: broadcast its key, payload and validity
                if (r < 4) { key[i] = quad_perm<0x00>(kc[comp]); val[i] = quad_perm<0x00>(vc[comp]); act[i] = quad_perm<0x00>(okc[comp]) != 0; }
                else if (r < 8) { key[i] = quad_perm<0x55>(kc[comp]); val[i] = quad_perm<0x55>(vc[comp]); act[i] = quad_perm<0x55>(okc[comp]) != 0; }
                else if (r < 12) { key[i] = quad_perm<0xAA>(kc[comp]); val[i] = quad_perm<0xAA>(vc[comp]); act[i] = quad_perm<0xAA>(okc[comp]) != 0; }
                else { key[i] = quad_perm<0xFF>(kc[comp]); val[i] = quad_perm<0xFF>(vc[comp]); act[i] = quad_perm<0xFF>(okc[comp]) != 0; }
                ln[i] = npj_bucket(key[i], factor, lines);
                q[i] = make_uint4(0, 0, 0, 0);
                if (act[i]) {
                    q[i] = t4[4 * ln[i] + sub];                         // 4 lanes x 16 bytes = the key's line
                }
            }
#pragma unroll
            for (int i = 0; i < B; ++i) {
                while (act[i]) {                                        // uniform inside the quad
                    // first empty bucket of the line, over the quad
                    uint32_t fe = q[i].x == 0u ? 2 * sub : (q[i].z == 0u ? 2 * sub + 1 : 8u);
                    fe = min(fe, quad_perm<0xB1>(fe));                  // lanes 0<->1, 2<->3
                    fe = min(fe, quad_perm<0x4E>(fe));                  // lanes 0<->2, 1<->3
                    const bool m0 = q[i].x == key[i] && 2 * sub < fe;
                    const bool m1 = q[i].z == key[i] && 2 * sub + 1 < fe;
                    const uint32_t m = (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
                    acc_n += m; acc_k += (u64)key[i] * m; acc_o += (u64)val[i] * m;
                    acc_i += (m0 ? q[i].y : 0u); acc_i += (m1 ? q[i].w : 0u);
                    if (MATERIALIZE) {
                        if (m0) em.emit(key[i], val[i], q[i].y);
                        if (m1) em.emit(key[i], val[i], q[i].w);
                    }
                    if (fe < 8u) break;                                 // the walk ends at the first empty bucket
                    if (++ln[i] == lines) ln[i] = 0;                    // full line: the walk goes on in the next one
                    q[i] = t4[4 * ln[i] + sub];
                }
            }
        }
    }
    if (MATERIALIZE && hj_lane() == 0)
        a.final_offsets[(u64)blockIdx.x * NW + wave] = wave_cursor[wave];
    acc_n = wave_reduce_sum(acc_n); acc_k = wave_reduce_sum(acc_k);
    acc_o = wave_reduce_sum(acc_o); acc_i = wave_reduce_sum(acc_i);
    if (hj_lane() == 0) { red[0][wave] = acc_n; red[1][wave] = acc_k; red[2][wave] = acc_o; red[3][wave] = acc_i; }
    __syncthreads();
    if (threadIdx.x < 4) {
        u64 s = 0;
        for (int i = 0; i < NW; ++i) s += red[threadIdx.x][i];
        if (s) atomicAdd(reinterpret_cast<u64 *>(a.result) + threadIdx.x, s);
    }
}

int hj_npj_probe_grid(int cus, size_t n)
{
    u64 blocks = ((n + 3) / 4 + NPJ_PROBE_BLOCK - 1) / NPJ_PROBE_BLOCK;
    if (blocks > (u64)cus * 8) blocks = (u64)cus * 8;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int hj_launch_npj_probe(const NpjProbeArgs &a, int cus, hipStream_t stream, int *grid_out)
{
    const int grid = hj_npj_probe_grid(cus, a.n);
    if (grid_out) *grid_out = grid;
    if (a.line_hash) {
        if (a.buckets % 8 != 0 || ((uintptr_t)a.table & 63)) return HJGPU_EINVAL;
        if (a.ok) hipLaunchKernelGGL(npj_probe_line_kernel<true>, dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
        else hipLaunchKernelGGL(npj_probe_line_kernel<false>, dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
        return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
    }
    const bool grouped = (a.buckets % 4 == 0) && (((uintptr_t)a.table & 31) == 0);
    if (grouped) hipLaunchKernelGGL(npj_probe_kernel<true>, dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
    else hipLaunchKernelGGL(npj_probe_kernel<false>, dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}

// --------------------------------------------------------------------------
// K9 close_gaps (npj.cpp:475-514).  Input: one end cursor per worker (wave);
// [cursor, end of its block) is a hole.  The filled region is made the dense
// prefix [0, J): tuples are taken from the highest filled positions and moved
// into the lowest holes.  Plan kernel (one workgroup): sort the holes (bitonic,
// LDS), then the reference's two-pointer walk (thread 0; <= 2*#workers steps)
// emits a move list; copy kernel: the whole chip executes the moves.  J is
// written to *dense_count.
// --------------------------------------------------------------------------
constexpr int CG_BLOCK = 1024;
constexpr int CG_MAX = 8192;      // max workers (waves) supported

struct Move { u64 dst, src, cnt; };

__global__ __launch_bounds__(CG_BLOCK) void close_gaps_plan_kernel(
    const u64 *__restrict__ final_offsets, uint32_t nworkers, u64 block_size,
    const u64 *__restrict__ block_counter, const uint32_t *__restrict__ overflow,
    Move *moves, uint32_t *nmoves, u64 *dense_count)
{
    // an overflowed join left cursors that do not describe disjoint holes: plan nothing
    if (*overflow) {
        if (threadIdx.x == 0) { *nmoves = 0; *dense_count = 0; }
        return;
    }
    __shared__ u64 hole_beg[CG_MAX];
    __shared__ u64 hole_end[CG_MAX];
    const int tid = threadIdx.x;
    // holes with no cursor sort to the end (key = ~0)
    for (uint32_t i = tid; i < CG_MAX; i += CG_BLOCK)
        hole_beg[i] = (i < nworkers) ? final_offsets[i] : HJ_NO_CURSOR;
    __syncthreads();
    for (uint32_t size = 2; size <= CG_MAX; size <<= 1) {
        for (uint32_t strd = size >> 1; strd > 0; strd >>= 1) {
            for (uint32_t i = tid; i < CG_MAX / 2; i += CG_BLOCK) {
                const uint32_t lo = 2 * i - (i & (strd - 1));
                const uint32_t hi = lo + strd;
                const bool up = ((lo & size) == 0);
                const u64 x = hole_beg[lo], y = hole_beg[hi];
                if ((x > y) == up) { hole_beg[lo] = y; hole_beg[hi] = x; }
            }
            __syncthreads();
        }
    }
    // block ends are fixed now; the walk below advances hole_beg only
    for (uint32_t i = tid; i < CG_MAX; i += CG_BLOCK)
        hole_end[i] = (hole_beg[i] == HJ_NO_CURSOR) ? HJ_NO_CURSOR
                                                    : (hole_beg[i] & ~(block_size - 1)) + block_size;
    __syncthreads();
    if (tid == 0) {
        uint32_t count = 0;
        while (count < nworkers && hole_beg[count] != HJ_NO_CURSOR) ++count;
        uint32_t nm = 0;
        u64 dense = 0;
        if (count) {
            // Two-pointer walk of npj.cpp:486-511 over the holes sorted by position.
            // The highest claimed block is some worker's last block, so it holds the
            // highest hole: nothing filled lies above hole_end[count-1].
            uint32_t l = 0, h = count - 1;
            u64 src = hole_end[h];
            uint32_t guard = 4 * count + 4;                 // the walk needs < 3*count steps
            while (l <= h && guard--) {
                const u64 fill = src - hole_end[h];         // filled tuples above hole h
                if (fill == 0) {
                    src = hole_beg[h];
                    if (h == 0) break;
                    --h;
                    continue;
                }
                const u64 hole = hole_end[l] - hole_beg[l];
                if (hole == 0) { ++l; continue; }
                const u64 cnt = fill < hole ? fill : hole;
                moves[nm].dst = hole_beg[l];
                moves[nm].src = src - cnt;
                moves[nm].cnt = cnt;
                ++nm;
                hole_beg[l] += cnt;
                src -= cnt;
            }
            dense = src;
        }
        *nmoves = nm;
        *dense_count = dense;
    }
}

// The whole chip copies the planned moves: one move per workgroup at a time
// (sources lie above every remaining hole, so a move never overlaps its
// destination or another move; a move is at most one block long).
__global__ __launch_bounds__(256) void close_gaps_copy_kernel(
    uint32_t *k, uint32_t *ov, uint32_t *iv, const Move *__restrict__ moves,
    const uint32_t *__restrict__ nmoves)
{
    const uint32_t nm = *nmoves;
    for (uint32_t m = blockIdx.x; m < nm; m += gridDim.x) {
        const u64 dst = moves[m].dst, src = moves[m].src, cnt = moves[m].cnt;
        for (u64 i = threadIdx.x; i < cnt; i += 256) {
            k[dst + i] = k[src + i];
            ov[dst + i] = ov[src + i];
            iv[dst + i] = iv[src + i];
        }
    }
}

int hj_launch_close_gaps_ex(uint32_t *k, uint32_t *ov, uint32_t *iv, const u64 *final_offsets,
                            uint32_t nworkers, u64 block_size, const u64 *block_counter,
                            const uint32_t *overflow, void *moves, uint32_t *nmoves,
                            u64 *dense_count, int cus, hipStream_t stream)
{
    if (nworkers > CG_MAX) return HJGPU_EINVAL;
    hipLaunchKernelGGL(close_gaps_plan_kernel, dim3(1), dim3(CG_BLOCK), 0, stream,
                       final_offsets, nworkers, block_size, block_counter, overflow,
                       reinterpret_cast<Move *>(moves), nmoves, dense_count);
    hipLaunchKernelGGL(close_gaps_copy_kernel, dim3(cus * 4), dim3(256), 0, stream, k, ov, iv,
                       reinterpret_cast<const Move *>(moves), nmoves);
    return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
}
