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
// UNIQUE: the reference's _UNIQUE build (npj.cpp:288-290, 436-438): the walk of a probe key ends at its first match.
template <bool GROUPED, bool UNIQUE>
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
                                if (UNIQUE) open = false;
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
                                if (UNIQUE) { act[j] = false; continue; }
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
        hj_store(&a.final_offsets[(u64)blockIdx.x * NW + wave], wave_cursor[wave]);
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
template <bool MATERIALIZE, bool UNIQUE>
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
                    bool m0 = q[i].x == key[i] && 2 * sub < fe;
                    bool m1 = q[i].z == key[i] && 2 * sub + 1 < fe;
                    bool found = false;                                 // UNIQUE: some lane of the quad holds a match
                    if (UNIQUE) {
                        // only the FIRST match of the walk counts: the lowest matching bucket of the line
                        uint32_t fm = m0 ? 2 * sub : (m1 ? 2 * sub + 1 : 8u);
                        const uint32_t mine = fm;
                        fm = min(fm, quad_perm<0xB1>(fm));
                        fm = min(fm, quad_perm<0x4E>(fm));
                        found = fm < 8u;
                        m0 = m0 && mine == fm && fm == 2 * sub;
                        m1 = m1 && mine == fm && fm == 2 * sub + 1;
                    }
                    const uint32_t m = (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
                    acc_n += m; acc_k += (u64)key[i] * m; acc_o += (u64)val[i] * m;
                    acc_i += (m0 ? q[i].y : 0u); acc_i += (m1 ? q[i].w : 0u);
                    if (MATERIALIZE) {
                        if (m0) em.emit(key[i], val[i], q[i].y);
                        if (m1) em.emit(key[i], val[i], q[i].w);
                    }
                    if (fe < 8u || (UNIQUE && found)) break;            // the walk ends at the first empty bucket (UNIQUE: first match)
                    if (++ln[i] == lines) ln[i] = 0;                    // full line: the walk goes on in the next one
                    q[i] = t4[4 * ln[i] + sub];
                }
            }
        }
    }
    if (MATERIALIZE && hj_lane() == 0)
        hj_store(&a.final_offsets[(u64)blockIdx.x * NW + wave], wave_cursor[wave]);
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
        if (a.ok && a.unique) hipLaunchKernelGGL((npj_probe_line_kernel<true, true>), dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
        else if (a.ok) hipLaunchKernelGGL((npj_probe_line_kernel<true, false>), dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
        else if (a.unique) hipLaunchKernelGGL((npj_probe_line_kernel<false, true>), dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((npj_probe_line_kernel<false, false>), dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
        return hipGetLastError() == hipSuccess ? HJGPU_OK : HJGPU_EHIP;
    }
    const bool grouped = (a.buckets % 4 == 0) && (((uintptr_t)a.table & 31) == 0);
    if (grouped && a.unique) hipLaunchKernelGGL((npj_probe_kernel<true, true>), dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
    else if (grouped) hipLaunchKernelGGL((npj_probe_kernel<true, false>), dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
    else if (a.unique) hipLaunchKernelGGL((npj_probe_kernel<false, true>), dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((npj_probe_kernel<false, false>), dim3(grid), dim3(NPJ_PROBE_BLOCK), 0, stream, a);
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

// Plan, in parallel.  Sort the holes by position (bitonic, LDS).  With top = end of the highest
// hole's block (the highest claimed block is some worker's last block: nothing filled lies above
// it), J = top - sum of the holes is the dense length.  The parts of the holes below J are the
// destinations; the filled stretches above J (between consecutive holes) are the sources; both
// lists are in position order, and slot t of one is paired with slot t of the other (the reference
// pairs the lowest holes with the HIGHEST tuples, npj.cpp:486-511: same set of rows in [0, J), and
// the order of a join result is unspecified).  Two prefix sums and, per hole, a binary search in
// the sources' prefix give the moves; a first pass counts them, a second writes them.
// (The two-pointer walk by one thread took 1.5 of close_gaps' 1.8 ms at 4096 workers.)
__global__ __launch_bounds__(CG_BLOCK) void close_gaps_plan_kernel(
    const u64 *__restrict__ final_offsets, uint32_t nworkers, u64 block_size,
    const u64 *__restrict__ block_counter, const uint32_t *__restrict__ overflow,
    Move *moves, uint32_t *nmoves, u64 *dense_count)
{
    // an overflowed join left cursors that do not describe disjoint holes: plan nothing
    if (*overflow) {
        if (threadIdx.x == 0) { hj_store(nmoves, 0u); hj_store(dense_count, (u64)0); }
        return;
    }
    __shared__ u64 hole_beg[CG_MAX];            // sorted hole starts, then prefix of the destination sizes
    __shared__ u64 hole_end[CG_MAX];            // prefix of the source sizes
    __shared__ u64 scratch[CG_BLOCK / 64 + 1];
    __shared__ u64 sh_count, sh_top, sh_holes;
    const int tid = threadIdx.x;
    uint32_t N = 64;                            // sort size: next power of two >= nworkers
    while (N < nworkers) N <<= 1;
    // holes with no cursor sort to the end (key = ~0)
    for (uint32_t i = tid; i < N; i += CG_BLOCK)
        hole_beg[i] = (i < nworkers) ? final_offsets[i] : HJ_NO_CURSOR;
    __syncthreads();
    for (uint32_t size = 2; size <= N; size <<= 1) {
        for (uint32_t strd = size >> 1; strd > 0; strd >>= 1) {
            for (uint32_t i = tid; i < N / 2; i += CG_BLOCK) {
                const uint32_t lo = 2 * i - (i & (strd - 1));
                const uint32_t hi = lo + strd;
                const bool up = ((lo & size) == 0);
                const u64 x = hole_beg[lo], y = hole_beg[hi];
                if ((x > y) == up) { hole_beg[lo] = y; hole_beg[hi] = x; }
            }
            __syncthreads();
        }
    }
    // number of real holes, top, total hole size (each thread owns the elements i = tid * per + j)
    const uint32_t per = (N + CG_BLOCK - 1) / CG_BLOCK;
    const uint32_t lo = min(N, (uint32_t)tid * per), hi = min(N, lo + per);
    auto end_of = [&](u64 b) -> u64 { return (b & ~(block_size - 1)) + block_size; };
    {
        u64 cnt = 0, holes = 0, top = 0;
        for (uint32_t i = lo; i < hi; ++i) {
            const u64 b = hole_beg[i];
            if (b != HJ_NO_CURSOR) { ++cnt; holes += end_of(b) - b; top = max(top, end_of(b)); }
        }
        if (tid == 0) { sh_count = 0; sh_top = 0; sh_holes = 0; }
        __syncthreads();
        if (cnt) { atomicAdd(&sh_count, cnt); atomicAdd(&sh_holes, holes); atomicMax(&sh_top, top); }
        __syncthreads();
    }
    const uint32_t count = (uint32_t)sh_count;
    const u64 top = sh_top, J = top - sh_holes;
    if (count == 0) {
        if (tid == 0) { hj_store(nmoves, 0u); hj_store(dense_count, (u64)0); }
        return;
    }
    // destination part of hole i: [b_i, min(e_i, J)); source stretch i: [max(e_{i-1}, J), b_i), i >= 1
    u64 dsz[CG_MAX / CG_BLOCK], ssz[CG_MAX / CG_BLOCK], sb[CG_MAX / CG_BLOCK], db[CG_MAX / CG_BLOCK];
    u64 dsum = 0, ssum = 0;
    for (uint32_t i = lo, j = 0; i < hi; ++i, ++j) {
        dsz[j] = ssz[j] = sb[j] = db[j] = 0;
        if (i < count) {
            const u64 b = hole_beg[i], e = end_of(b);
            db[j] = b;
            if (b < J) dsz[j] = min(e, J) - b;
            const u64 from = max(i > 0 ? end_of(hole_beg[i - 1]) : 0ull, J);    // filled from here up to the hole
            if (b > from) { sb[j] = from; ssz[j] = b - from; }
            dsum += dsz[j]; ssum += ssz[j];
        }
    }
    __syncthreads();                             // everybody has read its neighbours' hole_beg
    u64 drun = block_exclusive_scan<CG_BLOCK, u64>(dsum, scratch);
    __syncthreads();
    u64 srun = block_exclusive_scan<CG_BLOCK, u64>(ssum, scratch);
    // the sorted starts are no longer needed in LDS: hole_beg := prefix of the destination sizes,
    // hole_end := prefix of the source sizes (the end of the last stretch is never needed)
    __syncthreads();
    for (uint32_t i = lo, j = 0; i < hi; ++i, ++j) {
        hole_beg[i] = drun; hole_end[i] = srun;
        drun += dsz[j]; srun += ssz[j];
    }
    __syncthreads();
    // hole i's destination slots are [D_i, D_i + dsz_i); source stretch s covers slots [S_s, S_s + ssz_s).
    // Count the moves of my holes: one per source stretch that overlaps the hole's slot interval.
    auto first_stretch = [&](u64 t) -> uint32_t {          // largest s with S_s <= t (stretches of size 0 share a prefix)
        uint32_t a = 0, b = count;
        while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (hole_end[m] <= t) a = m; else b = m; }
        return a;
    };
    // the stretches' start positions are needed by other threads and LDS is full: global scratch
    // behind the move list (same workgroup, read after a barrier)
    u64 *stretch_beg = reinterpret_cast<u64 *>(moves + 2 * HJ_MAX_WORKERS);
    for (uint32_t i = lo, j = 0; i < hi; ++i, ++j) if (i < count) hj_store(&stretch_beg[i], sb[j]);
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = lo, j = 0; i < hi; ++i, ++j) {
        if (i >= count || dsz[j] == 0) continue;
        const u64 t0 = hole_beg[i], t1 = t0 + dsz[j];
        uint32_t s = first_stretch(t0);
        while (s < count) {
            const u64 s0 = hole_end[s], s1 = (s + 1 < count) ? hole_end[s + 1] : ~0ull;
            if (s0 >= t1) break;
            if (s1 > t0 && s1 > s0) ++mine;
            ++s;
        }
    }
    __syncthreads();
    const u64 mbase = block_exclusive_scan<CG_BLOCK, u64>((u64)mine, scratch);
    uint32_t at = (uint32_t)mbase;
    for (uint32_t i = lo, j = 0; i < hi; ++i, ++j) {
        if (i >= count || dsz[j] == 0) continue;
        const u64 t0 = hole_beg[i], t1 = t0 + dsz[j];
        uint32_t s = first_stretch(t0);
        while (s < count) {
            const u64 s0 = hole_end[s], s1 = (s + 1 < count) ? hole_end[s + 1] : ~0ull;
            if (s0 >= t1) break;
            if (s1 > t0 && s1 > s0) {
                const u64 a0 = max(t0, s0), a1 = min(t1, s1);       // overlapping slots
                hj_store(&moves[at].dst, db[j] + (a0 - t0));
                hj_store(&moves[at].src, stretch_beg[s] + (a0 - s0));
                hj_store(&moves[at].cnt, a1 - a0);
                ++at;
            }
            ++s;
        }
    }
    if (tid == CG_BLOCK - 1) hj_store(nmoves, (uint32_t)(mbase + mine));
    if (tid == 0) hj_store(dense_count, J);
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
            // (non-temporal like the rows themselves: a move that is lost leaves a hole's stale row inside the dense result)
            hj_store(&k[dst + i], k[src + i]);
            hj_store(&ov[dst + i], ov[src + i]);
            hj_store(&iv[dst + i], iv[src + i]);
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
