// ubench_partial_line_race.hip — are PLAIN partial-line stores lost (or read stale) when ANOTHER queue's kernels start and end
// beside the writing kernel?  No library code: the store pattern of K6 and nothing else.
//
// Round 5 found K6 losing tuples in 1.5 of 10^4 steps of the multi-stream slice pipeline with plain stores and in none with
// non-temporal stores.  What K6 does that round 4's stand-alone kernel (ubench_scratch_race.hip: whole
// 16-byte-per-lane lines, one wave per line) did not: its output RUNS end inside 128-byte lines, and the rest of such a line
// belongs to the run of another workgroup - usually on another XCD, written at another time.  So here:
//   * the output (SLOTS 8-byte slots) is cut into runs of 1 ... 48 slots; run k belongs to tile hash(k) % TILES; tiles are
//     claimed from a ticket counter by a persistent grid of 1024-thread workgroups (one per CU); a 16-lane group writes a run
//     like K6 does - 16-byte stores for aligned slot pairs inside the run, 8-byte stores for the ragged ends;
//   * every iteration writes value(iteration, slot) into the SAME buffer, so a slot that keeps value(iteration - 1, slot) was
//     either never written (lost store) or is read stale;
//   * a checker kernel on the writer's stream counts slots != value(iteration, slot) right behind the writer;
//   * meanwhile a second stream (and optionally a third) starts and ends small kernels, a streaming kernel and small
//     device-to-device copies: foreign kernel boundaries (every one an L2 write-back + invalidate by the command processor).
// On a mismatch the device is synchronised, the buffer is copied to the host with hipMemcpy (SDMA: not through any XCD's L2)
// and counted there, and a FRESH checker kernel counts again: memory wrong on the host = the store was lost; memory right and
// only the first kernel wrong = a stale read.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_partial_line_race.hip -o hash_join_codes_knl_amd/lib/ubench_partial_line_race
//   usage: ubench_partial_line_race [seconds per mode = 20] [mode = all]      modes: see `modes` below
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
typedef unsigned long long u64;

#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int BLOCK = 1024;
constexpr u64 SLOTS = 64ull << 20;          // 512 MiB of 8-byte slots
constexpr uint32_t TILES = 4096;

struct Report { u64 bad, stale, first[8][2]; };

__host__ __device__ inline u64 value_of(uint32_t iter, u64 slot) { return ((u64)iter << 40) | slot; }
__host__ __device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }

template <int STORE>            // 0 plain, 1 non-temporal, 2 only the 8-byte (partial-line) stores non-temporal
__device__ __forceinline__ void store16(u64 *p, u64 a, u64 b)
{
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    v4u_t t = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    if constexpr (STORE == 1) __builtin_nontemporal_store(t, reinterpret_cast<v4u_t *>(p));
    else *reinterpret_cast<v4u_t *>(p) = t;
}
template <int STORE>
__device__ __forceinline__ void store8(u64 *p, u64 a)
{
    if constexpr (STORE != 0) __builtin_nontemporal_store(a, p); else *p = a;
}

// K6's barrier: LDS traffic only, stores stay in flight across it and across the end of the wave
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// run k = slots [run_start[k], run_start[k + 1]); the runs of tile t are run_of_tile[tile_first[t] ... tile_first[t + 1])
template <int STORE, bool PRIVATE>
__global__ __launch_bounds__(BLOCK) void writer_kernel(u64 *__restrict__ out, const u64 *__restrict__ run_start, const uint32_t *__restrict__ run_of_tile,
                                                       const uint32_t *__restrict__ tile_first, uint32_t iter, uint32_t *ticket, uint32_t *sink)
{
    extern __shared__ uint32_t lds[];            // 150 KiB asked for: one workgroup per CU, like K6
    __shared__ uint32_t next;
    volatile uint32_t priv[2];                   // PRIVATE: one private word, written once (the "amplifier" of rounds 3-4)
    if (PRIVATE) { priv[0] = iter; }
    const uint32_t gid = threadIdx.x >> 4, sub = threadIdx.x & 15;
    for (;;) {
        if (threadIdx.x == 0) next = atomicAdd(ticket, 1u);
        lds_barrier();
        const uint32_t t = next;
        lds_barrier();
        if (t >= TILES) break;
        const uint32_t r0 = tile_first[t], r1 = tile_first[t + 1];
        for (uint32_t r = r0 + gid; r < r1; r += BLOCK / 16) {
            const uint32_t k = run_of_tile[r];
            const u64 b = run_start[k], e = run_start[k + 1];
            // the group walks the run in 128-byte lines of the output; lane `sub` owns the 16-byte slot pair `sub` of a line
            for (u64 s = (b & ~15ull) + 2 * sub; s < e; s += 32) {
                const bool v0 = s >= b, v1 = s + 1 >= b && s + 1 < e;
                if (v0 && v1) store16<STORE>(out + s, value_of(iter, s), value_of(iter, s + 1));
                else if (v0) store8<STORE>(out + s, value_of(iter, s));
                else if (v1) store8<STORE>(out + s + 1, value_of(iter, s + 1));
            }
        }
    }
    if (PRIVATE && priv[0] == 0xFFFFFFFFu) *sink = 1;
}

__global__ __launch_bounds__(BLOCK) void checker_kernel(const u64 *__restrict__ out, uint32_t iter, Report *rep)
{
    u64 bad = 0, stale = 0;
    for (u64 s = (u64)blockIdx.x * BLOCK + threadIdx.x; s < SLOTS; s += (u64)gridDim.x * BLOCK) {
        const u64 v = out[s];
        if (v != value_of(iter, s)) {
            ++bad;
            if (v == value_of(iter - 1, s)) ++stale;
            const u64 at = atomicAdd(&rep->bad, 1ull);
            if (at < 8) { rep->first[at][0] = s; rep->first[at][1] = v; }
        }
    }
    if (stale) atomicAdd(&rep->stale, stale);
    (void)bad;
}

__global__ void zero_kernel(uint32_t *p) { *p = 0; }
__global__ void tick_kernel(uint32_t *p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
// a neighbour that really uses the memory system: streams `n` 16-byte vectors
__global__ __launch_bounds__(512) void stream_kernel(const uint4 *__restrict__ in, u64 n, uint4 *sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u64 i = (u64)blockIdx.x * 512 + threadIdx.x; i < n; i += (u64)gridDim.x * 512) { const uint4 v = in[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    if (acc.x == 0x12345678u && acc.y == 17) *sink = acc;
}
// a neighbour that writes: a copy (the exchange's self-message, the join's row stores)
__global__ __launch_bounds__(512) void copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n)
{
    for (u64 i = (u64)blockIdx.x * 512 + threadIdx.x; i < n; i += (u64)gridDim.x * 512) out[i] = in[i];
}

template <int STORE, bool PRIVATE>
static void launch_writer_t(int cus, size_t lds, hipStream_t st, u64 *out, const u64 *runs, const uint32_t *rot, const uint32_t *tf, uint32_t iter, uint32_t *ticket, uint32_t *sink)
{
    hipLaunchKernelGGL((writer_kernel<STORE, PRIVATE>), dim3(cus), dim3(BLOCK), lds, st, out, runs, rot, tf, iter, ticket, sink);
}
static void launch_writer(int store, bool priv, int cus, size_t lds, hipStream_t st, u64 *out, const u64 *runs, const uint32_t *rot, const uint32_t *tf, uint32_t iter, uint32_t *ticket, uint32_t *sink)
{
    if (store == 0 && !priv) launch_writer_t<0, false>(cus, lds, st, out, runs, rot, tf, iter, ticket, sink);
    if (store == 0 && priv) launch_writer_t<0, true>(cus, lds, st, out, runs, rot, tf, iter, ticket, sink);
    if (store == 1 && !priv) launch_writer_t<1, false>(cus, lds, st, out, runs, rot, tf, iter, ticket, sink);
    if (store == 1 && priv) launch_writer_t<1, true>(cus, lds, st, out, runs, rot, tf, iter, ticket, sink);
    if (store == 2 && !priv) launch_writer_t<2, false>(cus, lds, st, out, runs, rot, tf, iter, ticket, sink);
    if (store == 2 && priv) launch_writer_t<2, true>(cus, lds, st, out, runs, rot, tf, iter, ticket, sink);
}

struct Mode { const char *name; int store; bool priv; int neighbours; bool flat_priority; bool deps; };
static const Mode modes[] = {
    {"plain stores, alone on the device", 0, false, 0, false, false},
    {"plain stores, tiny kernels + stream + copies on a second stream (priorities apart)", 0, false, 1, false, false},
    {"plain stores, the same on two more streams, ONE priority", 0, false, 2, true, false},
    {"plain stores + one private word in the writer, neighbours on two streams", 0, true, 2, true, false},
    {"non-temporal partial-line stores only (the product's solo K6), neighbours on two streams", 2, false, 2, true, false},
    {"all stores non-temporal (the product's default K6), neighbours on two streams", 1, false, 2, true, false},
    // what the library's pipelines have beside K6 and the modes above lack: kernels that WAIT FOR EVENTS OF ANOTHER QUEUE (their
    // dispatch carries a wider acquire), copies from and to page-locked HOST memory in both directions, host-side event waits
    {"plain stores; neighbours chained across two streams by events, with host <-> device copies", 0, false, 2, true, true},
    {"plain stores + private word; the same neighbours", 0, true, 2, true, true},
    {"all stores non-temporal; the same neighbours", 1, false, 2, true, true},
};

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s (%s), %d CUs; %llu slots (%llu MiB), %u tiles; %.0f s per mode\n", prop.name, prop.gcnArchName, cus, SLOTS, SLOTS * 8 >> 20, TILES, seconds);

    // the runs and their owners (host, once)
    std::vector<u64> run_start;
    { u64 s = 0; uint32_t k = 0; while (s < SLOTS) { run_start.push_back(s); s += 1 + mix(k * 2 + 1) % 48; ++k; } run_start.push_back(SLOTS); }
    const uint32_t runs = (uint32_t)run_start.size() - 1;
    std::vector<uint32_t> tile_first(TILES + 1, 0), run_of_tile(runs);
    for (uint32_t k = 0; k < runs; ++k) ++tile_first[mix(k * 2) % TILES + 1];
    for (uint32_t t = 0; t < TILES; ++t) tile_first[t + 1] += tile_first[t];
    { std::vector<uint32_t> at(tile_first.begin(), tile_first.end() - 1); for (uint32_t k = 0; k < runs; ++k) run_of_tile[at[mix(k * 2) % TILES]++] = k; }
    u64 shared_lines = 0;
    for (uint32_t k = 1; k < runs; ++k) if (run_start[k] & 15) ++shared_lines;
    printf("# %u runs, %llu of %llu lines are shared by the runs of two different tiles\n", runs, shared_lines, SLOTS / 16);

    u64 *d_out, *d_runs; uint32_t *d_rot, *d_tf, *d_ticket, *d_sink, *d_ticks; Report *d_rep; uint4 *d_nb, *d_nb2;
    const u64 nb_vecs = (256ull << 20) / 16;
    CHECK(hipMalloc(&d_out, SLOTS * 8 + 256)); CHECK(hipMalloc(&d_runs, run_start.size() * 8)); CHECK(hipMalloc(&d_rot, runs * 4));
    CHECK(hipMalloc(&d_tf, (TILES + 1) * 4)); CHECK(hipMalloc(&d_ticket, 256)); CHECK(hipMalloc(&d_sink, 256)); CHECK(hipMalloc(&d_ticks, 256));
    CHECK(hipMalloc(&d_rep, sizeof(Report))); CHECK(hipMalloc(&d_nb, nb_vecs * 16)); CHECK(hipMalloc(&d_nb2, nb_vecs * 16));
    CHECK(hipMemcpy(d_runs, run_start.data(), run_start.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rot, run_of_tile.data(), runs * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tf, tile_first.data(), (TILES + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_out, 0, SLOTS * 8)); CHECK(hipMemset(d_nb, 1, nb_vecs * 16)); CHECK(hipMemset(d_ticks, 0, 256));
    std::vector<u64> h_out(SLOTS);
    void *h_pin = nullptr;
    CHECK(hipHostMalloc(&h_pin, 4 << 20, hipHostMallocDefault));
    memset(h_pin, 7, 4 << 20);
    hipEvent_t ev_dep[2];
    for (hipEvent_t &e : ev_dep) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const size_t lds = 150 * 1024;
    const void *writers[3][2] = {{(const void *)writer_kernel<0, false>, (const void *)writer_kernel<0, true>},
                                 {(const void *)writer_kernel<1, false>, (const void *)writer_kernel<1, true>},
                                 {(const void *)writer_kernel<2, false>, (const void *)writer_kernel<2, true>}};
    for (auto &row : writers) for (const void *k : row) CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    uint32_t iter = 1;
    int total_bad_modes = 0;
    for (int m = 0; m < (int)(sizeof(modes) / sizeof(modes[0])); ++m) {
        if (only >= 0 && m != only) continue;
        const Mode &md = modes[m];
        hipStream_t sw, sn[2];
        CHECK(hipStreamCreateWithPriority(&sw, hipStreamNonBlocking, 0));
        for (int i = 0; i < 2; ++i) CHECK(hipStreamCreateWithPriority(&sn[i], hipStreamNonBlocking, md.flat_priority ? 0 : (i == 0 ? greatest : least)));
        u64 iterations = 0, boundaries = 0, bad_iterations = 0;
        const auto t0 = std::chrono::steady_clock::now();
        Report h;
        int reps = 1;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            // the neighbours' work: enqueued first, `reps` batches so that it lasts as long as the writer's launch (adapted below), never
            // waited for inside an iteration
            for (int i = 0; i < md.neighbours; ++i)
                for (int rep = 0; rep < reps; ++rep) {
                    for (int k = 0; k < 12; ++k) { hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, sn[i], d_ticks + 16 * i); ++boundaries; }
                    hipLaunchKernelGGL(stream_kernel, dim3(cus), dim3(512), 0, sn[i], d_nb, nb_vecs / 8, (uint4 *)d_sink + 4); ++boundaries;
                    hipLaunchKernelGGL(copy_kernel, dim3(cus / 2), dim3(512), 0, sn[i], d_nb, d_nb2, nb_vecs / 16); ++boundaries;
                    CHECK(hipMemcpyAsync(d_nb2 + (nb_vecs / 2), d_nb, 1 << 20, hipMemcpyDeviceToDevice, sn[i])); ++boundaries;
                    if (md.deps) {
                        // host -> device, a kernel behind it, an event; the OTHER neighbour stream waits for that event, runs a kernel, copies
                        // device -> host: every kernel here starts behind another queue's work or a copy engine's
                        const int o = 1 - i;
                        CHECK(hipMemcpyAsync(d_nb2 + (nb_vecs / 4) * (size_t)(i + 1), h_pin, 1 << 20, hipMemcpyHostToDevice, sn[i])); ++boundaries;
                        hipLaunchKernelGGL(copy_kernel, dim3(cus / 4), dim3(512), 0, sn[i], d_nb2 + (nb_vecs / 4) * (size_t)(i + 1), d_nb2, (u64)(1 << 16)); ++boundaries;
                        CHECK(hipEventRecord(ev_dep[i], sn[i]));
                        CHECK(hipStreamWaitEvent(sn[o], ev_dep[i], 0));
                        hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, sn[o], d_ticks + 16 * o); ++boundaries;
                        CHECK(hipMemcpyAsync((char *)h_pin + (2 << 20) + (o << 19), d_nb2, 1 << 19, hipMemcpyDeviceToHost, sn[o])); ++boundaries;
                    }
                }
            // (the pipelines' K6 launches start behind events of other streams - the exchange's arrival -: so does this writer)
            if (md.deps && md.neighbours) CHECK(hipStreamWaitEvent(sw, ev_dep[0], 0));
            hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(1), 0, sw, d_ticket);
            CHECK(hipMemsetAsync(d_rep, 0, sizeof(Report), sw));
            launch_writer(md.store, md.priv, cus, lds, sw, d_out, d_runs, d_rot, d_tf, iter, d_ticket, d_sink);
            hipLaunchKernelGGL(checker_kernel, dim3(cus * 2), dim3(BLOCK), 0, sw, d_out, iter, d_rep);
            CHECK(hipMemcpyAsync(&h, d_rep, sizeof(Report), hipMemcpyDeviceToHost, sw));
            CHECK(hipStreamSynchronize(sw));
            ++iterations;
            if (md.neighbours) {
                // neighbours idle when the writer's check is done: more of them next time; a backlog is drained every 32 iterations
                if (hipStreamQuery(sn[0]) == hipSuccess) { if (reps < 64) ++reps; } else (void)hipGetLastError();
                if (iterations % 32 == 0) { for (int i = 0; i < md.neighbours; ++i) CHECK(hipStreamSynchronize(sn[i])); }
            }
            if (h.bad) {
                ++bad_iterations;
                if (bad_iterations <= 4) {
                    printf("mode %d iteration %llu: checker behind the writer: %llu slots wrong, %llu of them hold the PREVIOUS iteration's value; first: slot %llu (slot %% 16 = %llu) holds %016llx\n",
                           m, iterations, h.bad, h.stale, h.first[0][0], h.first[0][0] % 16, h.first[0][1]);
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipMemcpy(h_out.data(), d_out, SLOTS * 8, hipMemcpyDeviceToHost));
                    u64 host_bad = 0, host_stale = 0, in_shared = 0;
                    for (u64 s = 0; s < SLOTS; ++s)
                        if (h_out[s] != value_of(iter, s)) { ++host_bad; if (h_out[s] == value_of(iter - 1, s)) ++host_stale; }
                    (void)in_shared;
                    Report h2;
                    CHECK(hipMemsetAsync(d_rep, 0, sizeof(Report), sw));
                    hipLaunchKernelGGL(checker_kernel, dim3(cus * 2), dim3(BLOCK), 0, sw, d_out, iter, d_rep);
                    CHECK(hipMemcpyAsync(&h2, d_rep, sizeof(Report), hipMemcpyDeviceToHost, sw));
                    CHECK(hipStreamSynchronize(sw));
                    printf("    device quiet: hipMemcpy to the host sees %llu slots wrong (%llu previous values); a fresh checker kernel sees %llu  ->  %s\n",
                           host_bad, host_stale, h2.bad, host_bad ? "the stores are LOST in memory" : "memory is right: the first checker read STALE data");
                    // lengths of the damaged stretches and where they sit in their lines
                    if (host_bad) {
                        u64 shown = 0;
                        for (u64 s = 0; s < SLOTS && shown < 6; ++s)
                            if (h_out[s] != value_of(iter, s)) { u64 e = s; while (e < SLOTS && h_out[e] != value_of(iter, e)) ++e; printf("    slots [%llu, %llu) (%llu slots, first at %llu of its line)\n", s, e, e - s, s % 16); s = e; ++shown; }
                    }
                }
            }
            ++iter;
        }
        CHECK(hipDeviceSynchronize());
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("mode %d (%s): %llu writer launches (%.2f ms each incl. check), %llu foreign kernel / copy boundaries beside them, %llu launches with WRONG output (neighbour batches per launch at the end: %d)\n",
               m, md.name, iterations, el * 1e3 / (double)iterations, boundaries, bad_iterations, reps);
        fflush(stdout);
        if (bad_iterations) ++total_bad_modes;
        CHECK(hipStreamDestroy(sw)); for (int i = 0; i < 2; ++i) CHECK(hipStreamDestroy(sn[i]));
    }
    return 0;
}
