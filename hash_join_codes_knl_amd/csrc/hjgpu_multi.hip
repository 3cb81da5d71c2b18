// hjgpu_multi.hip — the multi-GPU joins of include/hjgpu.h: communicators, transports, orchestration.
//
// The reference's exchange between workers is part of run_hj: phj.cpp:1715-1770 (thread-level pass: histogram_shared
// / interleave / partition_shared move BOTH relations between the threads) and cpra2.cpp:1861-1971 (thread t owns
// partitions [t*P/T, (t+1)*P/T), 1868-1872, and gathers them from every thread's chunk by memcpy, 1891-1904 and
// 1946-1959, after the counts were published behind a barrier, 1834-1840).  Here a worker is a GPU:
//   * PHJ / NPJ replicate the build side and shard the probe side (R join S = union_g R join S_g): one exchange,
//     overlapped with the probe side's partitioning through hjgpu_phj_overlapped_async's event.
//   * CPRA keeps the reference's shape: own-chunk partitioning with fan-out G, counts all-gather, all-to-all-v,
//     local PHJ.  The probe side travels in slices so that partitioning, transfer and join overlap.
// Everything above the Transport interface is transport-agnostic: ownership, counts, split sizes, slicing,
// reductions.  RcclTransport calls RCCL (xGMI); LoopbackTransport moves the same messages with hipMemcpyAsync
// between the ranks' buffers, so the whole orchestration runs at any world size on ONE GPU (tests).
// This file only uses the public C-ABI of the single-GPU library (hjgpu_partition_async, hjgpu_phj_build, ...).
//
// Barrier discipline.  The reference's workers meet at pthread barriers (cpra2.cpp:1834-1840, phj.cpp:1715-1770); a
// worker that never arrives hangs the program.  Here every host-side wait (sync_all, the counts the host needs in
// the middle of a CPRA step) honours the communicator's deadline (option "timeout_ms"): the streams are polled, RCCL
// is asked for asynchronous errors, and when the deadline passes the communicator is aborted (ncclCommAbort) and the
// call returns HJGPU_ERCCL with the rank and stream that did not finish - never a hang.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <chrono>
#include <memory>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

#include "../../include/hjgpu.h"
#include "exchange_layout.hpp"

// The library's store policy (csrc/hj_device.hpp): no plain global store beside other queues' work - K6's were lost in memory, 1.3-1.5 in
// 10^4 pipeline steps.  This file's clears and device-to-device copies are the library's own kernels (csrc/gen_kernels.hip), not the
// runtime's hipMemsetAsync / hipMemcpyAsync, and its few small kernels store non-temporally.
hipError_t hj_zero_async(void *p, size_t bytes, hipStream_t stream);
hipError_t hj_copy_async(void *dst, const void *src, size_t bytes, hipStream_t stream);

typedef unsigned long long u64;


namespace {

const uint32_t TOP_LEVEL_FACTOR = 0x2C1B3C6Du;        // odd multiplier of the exchange-level partitioning

// ---- RCCL, bound on first use ---------------------------------------------------------------------------------
// libhjgpu.so does not link librccl: single-GPU users (and the C-ABI tests, and hosts on a box without RCCL) load the
// library without it.  The first communicator that asks for the RCCL transport binds librccl.so.1 - the copy that is
// already in the process if there is one (torch ships its own), else the system's.
struct Rccl {
    void *lib = nullptr;
    char why[256] = {0};
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // already in the process
        for (const char *n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) {
            const char *root = getenv("ROCM_PATH");
            char path[512];
            snprintf(path, sizeof(path), "%s/lib/librccl.so.1", root && *root ? root : "/opt/rocm");
            r.lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) { snprintf(r.why, sizeof(r.why), "librccl.so.1 cannot be loaded: %s", dlerror()); return; }
        bool all = true;
#define BIND(name) do { r.name = reinterpret_cast<decltype(r.name)>(dlsym(r.lib, "nccl" #name)); if (!r.name) { all = false; snprintf(r.why, sizeof(r.why), "librccl lacks nccl" #name); } } while (0)
        BIND(GetVersion); BIND(GetUniqueId); BIND(CommInitAll); BIND(CommInitRank); BIND(CommDestroy); BIND(CommAbort);
        BIND(CommGetAsyncError); BIND(CommCount); BIND(CommUserRank); BIND(CommCuDevice); BIND(GroupStart); BIND(GroupEnd);
        BIND(Send); BIND(Recv); BIND(AllGather); BIND(Broadcast); BIND(AllReduce); BIND(GetErrorString);
#undef BIND
        if (!all) r.lib = nullptr;
    });
    return r.lib ? &r : nullptr;
}

// text of the last communicator that could not be created on this thread: hjgpu_comm_last_error(NULL)
thread_local char g_create_error[512] = "";

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

// One LOCAL rank: a GPU's share of the join.
struct Rank {
    int device = 0, global = 0;
    hjgpu_ctx *join = nullptr;      // local joins (keeps a prepared build side across the probe slices)
    hjgpu_ctx *part = nullptr;      // exchange-level partitioning plans in its own workspace
    hipStream_t main = nullptr;     // local joins
    hipStream_t comm = nullptr;     // exchanges
    hipStream_t prep = nullptr;     // exchange-level partitioning
    hipStream_t up = nullptr;       // hjgpu_join_host_multi: uploads of this rank's shard
    hipEvent_t ev_ready = nullptr;  // build side replicated (PHJ / NPJ)
    hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;      // timing of an exchange on `comm`
    std::vector<hipEvent_t> ev_w;                     // [2 * slice]: a slice's join before / after its wait for the exchange
    hipEvent_t ev_part[2] = {nullptr, nullptr};       // send buffers of slot b partitioned
    hipEvent_t ev_xchg[2] = {nullptr, nullptr};       // receive buffers of slot b filled
    hipEvent_t ev_join[2] = {nullptr, nullptr};       // receive buffers of slot b joined (free again)
    hipEvent_t ev_rx = nullptr;                       // build side received (CPRA)
    // fused counts (CpraStep::fused): per probe slot, this rank's fused histogram of its slice [fan-out * F2], everybody's
    // [G][fan-out * F2], and the rows of it that describe what this rank received, in the order of its pieces [G][k * F2]
    Buf cnt2[2], cnt2_all[2], cnt_recv[2];
    hipEvent_t ev_dbg = nullptr, ev_dbg2 = nullptr;   // option "debug_serialize"
    hipEvent_t ev_up_s = nullptr, ev_up_r = nullptr;  // host path: probe shard / build columns uploaded
    std::vector<hipEvent_t> ev_up_slice;              // host path, CPRA: probe slice i of the shard uploaded
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;      // host path: first kernel of the join started / upload finished (timing)
    hipEvent_t lb_in = nullptr, lb_out = nullptr;     // loopback transport
    Buf rbuf;                       // PHJ / NPJ: replicated build side (keys | payloads)
    Buf send_k[2], send_v[2], recv_k[2], recv_v[2];   // CPRA: probe-side slices, double-buffered
    Buf rsend_k, rsend_v, rrecv_k, rrecv_v;           // CPRA: build side
    u64 want_rows[3] = {0, 0, 0};   // CPRA, exchange in place: rows the send buffer (build side, probe slot 0 / 1) should hold next time
    Buf d_off;                      // [2][OFF_WORDS] u64: partition offsets of slot b
    Buf d_cnt;                      // [G] u64 send counts | [G * G] gathered matrix
    Buf d_res;                      // [12] u64: accumulated result | zero-key flag, overflow flag, 2 spare | last batch
    Buf scratch;                    // loopback all-reduce staging
    Buf pre;                        // preflight buffers (their own: the loopback all-reduce stages in `scratch`)
    Buf shard[4];                   // hjgpu_join_host_multi: this rank's share of ik, iv, ok, ov (kept between calls)
    Buf rows_col[3];                // hjgpu_join_host_rows_multi: this rank's result columns
    u64 *h_pin = nullptr;           // pinned host scratch, see hp_*()
};

struct Transport;

}  // namespace

struct hjgpu_comm {
    int nranks = 0, first = 0;
    std::vector<Rank> ranks;                 // local ranks
    std::unique_ptr<Transport> transport;
    bool ring_broadcast = false;
    size_t max_message_bytes = (size_t)1 << 30;
    int reserve_cus = -1;                    // -1: 16 with RCCL and more than one rank, else 0 (option "reserve_cus")
    int timeout_ms = 0;                      // deadline of every host-side wait; 0 = none (option "timeout_ms")
    bool broken = false;                     // a deadline expired / RCCL reported an asynchronous error: aborted
    int stall_rank = -1, stall_ms = 0;       // loopback fault injection (options "stall_rank", "stall_ms")
    bool self_via_rccl = false;              // option "self_via_rccl": a rank's message to itself goes through ncclSend / ncclRecv too (tests:
                                             // grouped point-to-point RCCL calls run on a one-GPU box that way)
    int debug_forensics = 0;                 // option "debug_forensics": option "audit" on every rank's two contexts; a CPRA step keeps the
                                             // records of its partitioning calls and joins (hjgpu_comm_get_forensics)
    std::vector<u64> forensics;              // per local rank: {global rank, partitioning records, join records}, then the records (32 words each)
    // "debug_forensics" = 2: every slice's records are read as soon as the slice's partitioning / join has finished - before the next use of
    // its buffers is enqueued - and a stage whose sums differ from its input's is looked at again ON THE SPOT (hjgpu_audit_recheck):
    // per event {global rank, context (0 partitioning, 1 join), slice, checks n, the call's record (32 words)} + n x 9 words
    std::vector<u64> frozen;
    int debug_serialize = 0;                 // option "debug_serialize" (diagnostics): bit 0 host waits after every slice's join, bit 1 the
                                             // partitioning waits for the joins enqueued so far, bit 2 the exchange waits for them,
                                             // bit 3 a join waits for the partitioning enqueued so far (with bit 1: the two never overlap)
    bool exchange_in_place = true;           // option "exchange_in_place": a CPRA rank keeps its own partitions where its partitioning wrote
                                             // them (last) and receives the other ranks' pieces right behind: the message to itself is never copied
    int cpra_k = 0;                          // option "cpra_k": partitions per rank of the exchange-level pass (0 = 192 / ranks); measurements:
                                             // k = 24 at world 1 gives the receiver the per-GPU work of an 8-GPU join
    bool host_rows_batched = true;           // option "host_rows_batched": hjgpu_join_host_rows_multi (PHJ / NPJ) = the one-GPU batched host pipeline per rank
    bool cpra_fused_counts = true;           // option "cpra_fused_counts": the senders' histogram pass counts the receivers' final partitions
                                             // too when G * k * F2 <= 32768 (build sides up to ~114 M rows): the receivers skip K4p
    bool cpra_two_level = false;             // option "cpra_two_level": round 2's CPRA (exchange with fan-out G, then a complete local PHJ)
    int cpra_grouped = 1;                    // option "cpra_grouped": a rank's share beyond two passes' reach is joined by a grouped plan (the ranks
                                             // agree on it from the relations' total sizes: one 16-byte all-reduce before the build side's exchange);
                                             // 1: where the grouped ROAD pays (cpra_join), 2: wherever hjgpu_grouped_plan groups (tests), 0: never
    char err[512];
    char why_broken[512];
    std::mutex err_mu;                       // the local ranks' enqueue work runs on one host thread per rank (each_rank)
    std::mutex abort_mu;                     // give_up(): one abort per communicator
};

namespace {

// pinned host scratch of a rank (u64 words)
constexpr size_t OFF_WORDS = 1040;           // partition offsets of one slot: fan-out <= 1024 (+ 1), or G + 1
inline size_t hp_words(size_t G) { return 2 * OFF_WORDS + G * G + 8 + G + 8; }
inline u64 *hp_off(const Rank &r, size_t G, int slot) { (void)G; return r.h_pin + (size_t)slot * OFF_WORDS; }  // [2][fan-out + 1] offsets
inline u64 *hp_matrix(const Rank &r, size_t G) { (void)G; return r.h_pin + 2 * OFF_WORDS; }                    // [G * G] counts matrix
inline u64 *hp_result(const Rank &r, size_t G) { return r.h_pin + 2 * OFF_WORDS + G * G; }                     // [8] global result + flags
inline u64 *hp_cnt(const Rank &r, size_t G) { return r.h_pin + 2 * OFF_WORDS + G * G + 8; }                    // [G] send counts
inline u64 *hp_local(const Rank &r, size_t G) { return r.h_pin + 2 * OFF_WORDS + G * G + 8 + G; }              // [8] this rank's own result (rows)

int cfail(hjgpu_comm *c, int status, const char *what, const char *detail = nullptr)
{
    if (c) {
        std::lock_guard<std::mutex> g(c->err_mu);
        snprintf(c->err, sizeof(c->err), detail ? "%s: %s" : "%s", what, detail);
    }
    return status;
}

// The enqueue work of the local ranks - about 25 kernel launches per rank and join - runs on one host thread per rank:
// issued from a single thread, rank 7 of an 8-GPU host started a millisecond after rank 0 on an 8 ms step (round 2).
// RCCL's grouped calls stay on the calling thread.  fn(l) returns an HJGPU_* status; the first failure is returned.
// (HJ_HOST_FORK / HJ_HOST_JOIN: nothing in the product; tests/cpp_pipeline_ordering.cpp, which runs this file on a CPU under a
// recorder of stream and event order, learns through them which host thread knows what about finished streams)
#ifndef HJ_HOST_FORK
#define HJ_HOST_FORK() ((void)0)
#define HJ_HOST_JOIN() ((void)0)
#endif
template <typename F>
int each_rank(int L, F fn)
{
    if (L == 1) return fn(0);
    std::vector<int> rc((size_t)L, HJGPU_OK);
    std::vector<std::thread> workers;
    HJ_HOST_FORK();
    for (int l = 1; l < L; ++l) workers.emplace_back([&rc, &fn, l] { rc[(size_t)l] = fn(l); });
    rc[0] = fn(0);
    for (std::thread &t : workers) t.join();
    HJ_HOST_JOIN();
    for (int l = 0; l < L; ++l) if (rc[(size_t)l] != HJGPU_OK) return rc[(size_t)l];
    return HJGPU_OK;
}

#define HIPM(c, call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return cfail((c), HJGPU_EHIP, #call, hipGetErrorString(e_));   \
    } while (0)
#define CHKM(call)                                                                           \
    do {                                                                                     \
        int s_ = (call);                                                                     \
        if (s_ != HJGPU_OK) return s_;                                                       \
    } while (0)
// a join-library call on rank `r`: its error text becomes the communicator's
#define JOINM(c, ctx, call)                                                                  \
    do {                                                                                     \
        int s_ = (call);                                                                     \
        if (s_ != HJGPU_OK) return cfail((c), s_, #call, hjgpu_last_error(ctx));             \
    } while (0)

int ensure(hjgpu_comm *c, const Rank &r, Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return HJGPU_OK;
    HIPM(c, hipSetDevice(r.device));
    if (b.p) { HIPM(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = (bytes + 4095) / 4096 * 4096 + 256;     // 16-byte tails for aligned vector reads
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return cfail(c, HJGPU_ENOMEM, "hipMalloc(exchange buffer)", hipGetErrorString(e)); }
    b.cap = want;
    return HJGPU_OK;
}

// The buffer the exchange-level partitioning scatters into (K6 pass 1 writes it through G * k frontiers): as sensitive to
// WHICH allocation it is as the join's own pass-1 twin (2.82 or 3.17 ms per 10^9 tuples) - it comes from
// the partition context's placement search (hjgpu_malloc_placed; plain allocation below 1 GiB).
int ensure_scatter_target(hjgpu_comm *c, const Rank &r, Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return HJGPU_OK;
    HIPM(c, hipSetDevice(r.device));
    if (b.p) { HIPM(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = (bytes + 4095) / 4096 * 4096 + 256;
    void *p = nullptr;
    JOINM(c, r.part, hjgpu_malloc_placed(r.part, &p, want));
    b.p = p; b.cap = want;
    return HJGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// Transport: collectives over ALL ranks, issued for all LOCAL ranks at once (arrays indexed by local rank).
// Every operation is enqueue-only on the given streams (one per local rank).
// ---------------------------------------------------------------------------------------------------
struct Transport {
    hjgpu_comm *c;
    explicit Transport(hjgpu_comm *comm) : c(comm) {}
    virtual ~Transport() {}
    int nlocal() const { return (int)c->ranks.size(); }
    virtual const char *name() const = 0;
    // recv[l][g * bytes .. ) = send of global rank g
    virtual int all_gather(const void *const *send, void *const *recv, size_t bytes, hipStream_t const *streams) = 0;
    // bufs[l] of `bytes` (capacity: replicate_capacity(bytes)) becomes the root's
    virtual int replicate(void *const *bufs, size_t bytes, int root, bool ring, hipStream_t const *streams) = 0;
    // element counts / offsets per peer: scnt[l][g] elements from send[l] + soff[l][g] go to rank g, which
    // stores them at recv + roff[g'][l'] ...; rcnt[l][g] must equal what rank g sends to local rank l
    virtual int all_to_all_v(const void *const *send, const u64 *const *soff, const u64 *const *scnt,
                             void *const *recv, const u64 *const *roff, const u64 *const *rcnt,
                             size_t elem_bytes, hipStream_t const *streams) = 0;
    // bufs[l][0..count) = sum over all ranks (uint64, wrap-around)
    virtual int all_reduce_u64(u64 *const *bufs, size_t count, hipStream_t const *streams) = 0;
    // an error RCCL met asynchronously (a peer died, a link failed): text in `what`, true = there is one
    virtual bool async_error(char *what, size_t n) { (void)what; (void)n; return false; }
    // give up: whatever the transport has in flight is cancelled, its resources are released without waiting
    virtual void abort() {}
    virtual void info(hjgpu_comm_info *i) { (void)i; }
    size_t replicate_capacity(size_t bytes) const
    {
        const size_t per = ((bytes + c->nranks - 1) / c->nranks + 15) & ~size_t(15);
        return per * c->nranks;
    }
};

// ---- RCCL over xGMI ---------------------------------------------------------------------------------
// ncclGroupStart ... ncclGroupEnd: the end is never skipped.  An early return from inside an open group would leave
// the thread in group mode and every later RCCL call - the ncclCommDestroy of the clean-up included - deferred.
struct Group {
    hjgpu_comm *c;
    Rccl *R;
    int rc = HJGPU_OK;
    bool open = false;
    Group(hjgpu_comm *comm, Rccl *r) : c(comm), R(r)
    {
        const ncclResult_t e = R->GroupStart();
        if (e != ncclSuccess) rc = cfail(c, HJGPU_ERCCL, "ncclGroupStart", R->GetErrorString(e));
        else open = true;
    }
    bool ok() const { return rc == HJGPU_OK; }
    void nccl(ncclResult_t e, const char *what) { if (e != ncclSuccess && rc == HJGPU_OK) rc = cfail(c, HJGPU_ERCCL, what, R->GetErrorString(e)); }
    void hip(hipError_t e, const char *what) { if (e != hipSuccess && rc == HJGPU_OK) rc = cfail(c, HJGPU_EHIP, what, hipGetErrorString(e)); }
    int end()
    {
        if (open) {
            open = false;
            const ncclResult_t e = R->GroupEnd();
            if (e != ncclSuccess && rc == HJGPU_OK) rc = cfail(c, HJGPU_ERCCL, "ncclGroupEnd", R->GetErrorString(e));
        }
        return rc;
    }
    ~Group() { if (open) (void)R->GroupEnd(); }
};

struct RcclTransport : Transport {
    Rccl *R;
    std::vector<ncclComm_t> comms;           // one per local rank
    RcclTransport(hjgpu_comm *comm, Rccl *r) : Transport(comm), R(r) {}
    ~RcclTransport() override
    {
        for (size_t l = 0; l < comms.size(); ++l)
            if (comms[l]) {
                (void)hipSetDevice(c->ranks[l].device);
                // a communicator that met an error is not waited for
                if (c->broken) (void)R->CommAbort(comms[l]); else (void)R->CommDestroy(comms[l]);
            }
    }
    const char *name() const override { return "rccl"; }

    bool async_error(char *what, size_t n) override
    {
        for (size_t l = 0; l < comms.size(); ++l) {
            ncclResult_t e = ncclSuccess;
            if (comms[l] && R->CommGetAsyncError(comms[l], &e) == ncclSuccess && e != ncclSuccess && e != ncclInProgress) {
                snprintf(what, n, "RCCL asynchronous error on rank %d: %s", c->ranks[l].global, R->GetErrorString(e));
                return true;
            }
        }
        return false;
    }

    void abort() override
    {
        for (size_t l = 0; l < comms.size(); ++l)
            if (comms[l]) { (void)hipSetDevice(c->ranks[l].device); (void)R->CommAbort(comms[l]); comms[l] = nullptr; }
    }

    void info(hjgpu_comm_info *i) override
    {
        int v = 0;
        if (R->GetVersion(&v) == ncclSuccess) i->rccl_version = v;
        if (!comms.empty() && comms[0]) {
            int n = -1, r = -1, d = -1;
            if (R->CommCount(comms[0], &n) == ncclSuccess) i->rccl_nranks = n;
            if (R->CommUserRank(comms[0], &r) == ncclSuccess) i->rccl_rank = r;
            if (R->CommCuDevice(comms[0], &d) == ncclSuccess) i->rccl_device = d;
        }
    }

    int all_gather(const void *const *send, void *const *recv, size_t bytes, hipStream_t const *streams) override
    {
        Group g(c, R);
        for (int l = 0; l < nlocal() && g.ok(); ++l) {
            g.hip(hipSetDevice(c->ranks[l].device), "hipSetDevice");
            if (g.ok()) g.nccl(R->AllGather(send[l], recv[l], bytes, ncclUint8, comms[l], streams[l]), "ncclAllGather");
        }
        return g.end();
    }

    int replicate(void *const *bufs, size_t bytes, int root, bool ring, hipStream_t const *streams) override
    {
        const int G = c->nranks;
        if (G == 1) return HJGPU_OK;
        if (ring || bytes < (size_t)G * 4096) {
            // one ring broadcast: bound by ONE xGMI link (512 MB of build side at ~60 GB/s per direction = 8.5 ms)
            Group g(c, R);
            for (int l = 0; l < nlocal() && g.ok(); ++l) {
                g.hip(hipSetDevice(c->ranks[l].device), "hipSetDevice");
                if (g.ok()) g.nccl(R->Broadcast(bufs[l], bufs[l], bytes, ncclUint8, root, comms[l], streams[l]), "ncclBroadcast");
            }
            return g.end();
        }
        // xGMI is point to point (7 links per GPU, every pair directly connected): the root sends a DIFFERENT
        // 1/G slice to every peer over its own link, then everybody exchanges slices - all links of all GPUs
        // carry data, 2 x ~1/(G-1) of the single-link time
        const size_t per = replicate_capacity(bytes) / G;
        {
            Group g(c, R);
            for (int l = 0; l < nlocal() && g.ok(); ++l) {
                const int me = c->ranks[l].global;
                g.hip(hipSetDevice(c->ranks[l].device), "hipSetDevice");
                char *b = static_cast<char *>(bufs[l]);
                if (me == root) {
                    for (int p = 0; p < G && g.ok(); ++p)
                        if (p != root) g.nccl(R->Send(b + (size_t)p * per, per, ncclUint8, p, comms[l], streams[l]), "ncclSend");
                } else if (g.ok()) g.nccl(R->Recv(b + (size_t)me * per, per, ncclUint8, root, comms[l], streams[l]), "ncclRecv");
            }
            CHKM(g.end());
        }
        Group g(c, R);
        for (int l = 0; l < nlocal() && g.ok(); ++l) {
            const int me = c->ranks[l].global;
            g.hip(hipSetDevice(c->ranks[l].device), "hipSetDevice");
            char *b = static_cast<char *>(bufs[l]);
            if (g.ok()) g.nccl(R->AllGather(b + (size_t)me * per, b, per, ncclUint8, comms[l], streams[l]), "ncclAllGather");   // in place
        }
        return g.end();
    }

    int all_to_all_v(const void *const *send, const u64 *const *soff, const u64 *const *scnt,
                     void *const *recv, const u64 *const *roff, const u64 *const *rcnt,
                     size_t elem_bytes, hipStream_t const *streams) override
    {
        // grouped point-to-point: every GPU pair has its own link, so all messages of a rank travel at once.
        // Messages are cut into pieces of at most max_message_bytes (both ends cut alike): a single 4 GB
        // message lost half its payload through torch's all_to_all_single on RCCL 2.26 in round 1.
        const int G = c->nranks;
        const u64 piece = c->max_message_bytes / elem_bytes ? c->max_message_bytes / elem_bytes : 1;
        // a rank's message to ITSELF (1 / G of every exchange) never touches a link: a device-to-device copy on the same
        // stream (5 TB/s of read + write) instead of RCCL's self send / receive (measured at world 1: 0.8 TB/s)
        const bool self_copy = !c->self_via_rccl;
        for (int l = 0; l < nlocal() && self_copy; ++l) {
            const int me = c->ranks[l].global;
            if (scnt[l][me] != rcnt[l][me]) return cfail(c, HJGPU_EINVAL, "all_to_all_v: a rank's counts for itself disagree");
            if (!scnt[l][me]) continue;
            HIPM(c, hipSetDevice(c->ranks[l].device));
            HIPM(c, hj_copy_async(static_cast<char *>(recv[l]) + roff[l][me] * elem_bytes,
                                   static_cast<const char *>(send[l]) + soff[l][me] * elem_bytes, scnt[l][me] * elem_bytes, streams[l]));
        }
        if (G == 1 && self_copy) return HJGPU_OK;
        Group g(c, R);
        for (int l = 0; l < nlocal() && g.ok(); ++l) {
            const int me = c->ranks[l].global;
            g.hip(hipSetDevice(c->ranks[l].device), "hipSetDevice");
            for (int p = 0; p < G && g.ok(); ++p) {
                if (p == me && self_copy) continue;
                const char *s = static_cast<const char *>(send[l]) + soff[l][p] * elem_bytes;
                for (u64 at = 0; at < scnt[l][p] && g.ok(); at += piece) {
                    const u64 n = scnt[l][p] - at < piece ? scnt[l][p] - at : piece;
                    g.nccl(R->Send(s + at * elem_bytes, n * elem_bytes, ncclUint8, p, comms[l], streams[l]), "ncclSend");
                }
                char *r = static_cast<char *>(recv[l]) + roff[l][p] * elem_bytes;
                for (u64 at = 0; at < rcnt[l][p] && g.ok(); at += piece) {
                    const u64 n = rcnt[l][p] - at < piece ? rcnt[l][p] - at : piece;
                    g.nccl(R->Recv(r + at * elem_bytes, n * elem_bytes, ncclUint8, p, comms[l], streams[l]), "ncclRecv");
                }
            }
        }
        return g.end();
    }

    int all_reduce_u64(u64 *const *bufs, size_t count, hipStream_t const *streams) override
    {
        Group g(c, R);
        for (int l = 0; l < nlocal() && g.ok(); ++l) {
            g.hip(hipSetDevice(c->ranks[l].device), "hipSetDevice");
            if (g.ok()) g.nccl(R->AllReduce(bufs[l], bufs[l], count, ncclUint64, ncclSum, comms[l], streams[l]), "ncclAllReduce");
        }
        return g.end();
    }
};

// ---- loopback: every rank in this process, messages are device-to-device copies -------------------------
__global__ void sum_rows_kernel(const u64 *__restrict__ rows, u64 *__restrict__ out, uint32_t nrows, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 s = 0;
    for (uint32_t r = 0; r < nrows; ++r) s += rows[(u64)r * count + i];
    __builtin_nontemporal_store(s, &out[i]);
}

// fault injection: a rank that does not arrive at a collective for `ms` (a host function on its stream: the GPU is
// idle meanwhile, nothing spins, and the stall ends by itself)
void stall_host_fn(void *arg)
{
    std::this_thread::sleep_for(std::chrono::milliseconds((long)(intptr_t)arg));
}

struct LoopbackTransport : Transport {
    explicit LoopbackTransport(hjgpu_comm *comm) : Transport(comm) {}
    const char *name() const override { return "loopback"; }
    struct Xfer { int src, dst; const void *from; void *to; size_t bytes; };

    // A collective: every rank's stream first reaches the call (lb_in), the copies run on the DESTINATION's
    // stream, and no stream goes on before all copies are done (lb_out), as after a collective that is
    // complete on the stream: send buffers may be rewritten, received data may be read.
    int run(const std::vector<Xfer> &xs, hipStream_t const *streams)
    {
        const int G = nlocal();
        for (int l = 0; l < G; ++l) {
            HIPM(c, hipSetDevice(c->ranks[l].device));
            if (l == c->stall_rank && c->stall_ms > 0)
                HIPM(c, hipLaunchHostFunc(streams[l], stall_host_fn, (void *)(intptr_t)c->stall_ms));
            HIPM(c, hipEventRecord(c->ranks[l].lb_in, streams[l]));
        }
        for (int d = 0; d < G; ++d) {
            HIPM(c, hipSetDevice(c->ranks[d].device));
            for (int s = 0; s < G; ++s)
                if (s != d) HIPM(c, hipStreamWaitEvent(streams[d], c->ranks[s].lb_in, 0));
            for (const Xfer &x : xs)
                if (x.dst == d && x.bytes) HIPM(c, hipMemcpyAsync(x.to, x.from, x.bytes, hipMemcpyDefault, streams[d]));
            HIPM(c, hipEventRecord(c->ranks[d].lb_out, streams[d]));
        }
        for (int l = 0; l < G; ++l) {
            HIPM(c, hipSetDevice(c->ranks[l].device));
            for (int q = 0; q < G; ++q)
                if (q != l) HIPM(c, hipStreamWaitEvent(streams[l], c->ranks[q].lb_out, 0));
        }
        return HJGPU_OK;
    }

    int all_gather(const void *const *send, void *const *recv, size_t bytes, hipStream_t const *streams) override
    {
        std::vector<Xfer> xs;
        for (int d = 0; d < nlocal(); ++d)
            for (int s = 0; s < nlocal(); ++s)
                xs.push_back({s, d, send[s], static_cast<char *>(recv[d]) + (size_t)s * bytes, bytes});
        return run(xs, streams);
    }

    int replicate(void *const *bufs, size_t bytes, int root, bool, hipStream_t const *streams) override
    {
        std::vector<Xfer> xs;
        for (int d = 0; d < nlocal(); ++d)
            if (d != root) xs.push_back({root, d, bufs[root], bufs[d], bytes});
        return run(xs, streams);
    }

    int all_to_all_v(const void *const *send, const u64 *const *soff, const u64 *const *scnt,
                     void *const *recv, const u64 *const *roff, const u64 *const *rcnt,
                     size_t elem_bytes, hipStream_t const *streams) override
    {
        std::vector<Xfer> xs;
        for (int s = 0; s < nlocal(); ++s)
            for (int d = 0; d < nlocal(); ++d) {
                if (scnt[s][d] != rcnt[d][s]) return cfail(c, HJGPU_EINVAL, "all_to_all_v: send and receive counts disagree");
                xs.push_back({s, d, static_cast<const char *>(send[s]) + soff[s][d] * elem_bytes,
                              static_cast<char *>(recv[d]) + roff[d][s] * elem_bytes, (size_t)(scnt[s][d] * elem_bytes)});
            }
        return run(xs, streams);
    }

    int all_reduce_u64(u64 *const *bufs, size_t count, hipStream_t const *streams) override
    {
        const int G = nlocal();
        std::vector<Xfer> xs;
        for (int d = 0; d < G; ++d) {
            CHKM(ensure(c, c->ranks[d], c->ranks[d].scratch, (size_t)G * count * sizeof(u64)));
            for (int s = 0; s < G; ++s)
                xs.push_back({s, d, bufs[s], static_cast<u64 *>(c->ranks[d].scratch.p) + (size_t)s * count, count * sizeof(u64)});
        }
        CHKM(run(xs, streams));           // nobody passes before every copy has read its source
        for (int d = 0; d < G; ++d) {
            HIPM(c, hipSetDevice(c->ranks[d].device));
            hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, streams[d],
                               static_cast<const u64 *>(c->ranks[d].scratch.p), bufs[d], (uint32_t)G, (uint32_t)count);
            HIPM(c, hipGetLastError());
        }
        return HJGPU_OK;
    }
};

// dst[0..3] += src[0..3]: the aggregates of one probe batch join the rank's running result
__global__ void add_result_kernel(u64 *__restrict__ dst, const u64 *__restrict__ src)
{
    if (threadIdx.x < 4) __builtin_nontemporal_store(dst[threadIdx.x] + src[threadIdx.x], &dst[threadIdx.x]);
}

// a status flag raised from the host side of the pipeline (stream-ordered with the joins that read / reduce it)
__global__ void bump_kernel(u64 *flag) { __builtin_nontemporal_store(*flag + 1, flag); }

// ---- construction ---------------------------------------------------------------------------------
int init_rank(hjgpu_comm *c, Rank &r, int device, int global)
{
    r.device = device; r.global = global;
    int rc = hjgpu_create(device, &r.join);
    if (rc != HJGPU_OK) return cfail(c, rc, "hjgpu_create(join context)");
    rc = hjgpu_create(device, &r.part);
    if (rc != HJGPU_OK) return cfail(c, rc, "hjgpu_create(partition context)");
    HIPM(c, hipSetDevice(device));
    // The runtime maps streams onto a few hardware queues PER PRIORITY (4 by default), and the commands of streams that
    // share a queue run in the order they were submitted - a wait of one stream holds back whatever another stream
    // submitted behind it.  With the two contexts' own streams a rank has more than four: the exchange and the upload
    // streams take the high priority, the exchange-level partitioning the low one, the joins stay normal - three queue
    // pools.  An exchange never waits behind a join that merely shares its queue (and RCCL's kernels find CUs ahead of
    // the next persistent grid), a join never behind the upload or the partitioning it is meant to overlap.
    int least = 0, greatest = 0;
    HIPM(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    // (diagnostics, read when the communicator is made: HJGPU_DEBUG_FLAT_PRIORITIES=1 puts all four streams at the default priority)
    const char *flat = getenv("HJGPU_DEBUG_FLAT_PRIORITIES");
    if (flat && flat[0] == '1') least = greatest = 0;
    HIPM(c, hipStreamCreateWithFlags(&r.main, hipStreamNonBlocking));
    HIPM(c, hipStreamCreateWithPriority(&r.prep, hipStreamNonBlocking, least));
    for (hipStream_t *s : {&r.comm, &r.up}) HIPM(c, hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest));
    hipEvent_t *timed[] = {&r.ev_x0, &r.ev_x1, &r.ev_t0, &r.ev_t1};
    for (hipEvent_t *e : timed) HIPM(c, hipEventCreate(e));
    hipEvent_t *plain[] = {&r.ev_ready, &r.ev_part[0], &r.ev_part[1], &r.ev_xchg[0], &r.ev_xchg[1],
                           &r.ev_join[0], &r.ev_join[1], &r.ev_rx, &r.ev_dbg, &r.ev_dbg2, &r.ev_up_s, &r.ev_up_r, &r.lb_in, &r.lb_out};
    for (hipEvent_t *e : plain) HIPM(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    const size_t G = (size_t)c->nranks;
    HIPM(c, hipHostMalloc(reinterpret_cast<void **>(&r.h_pin), hp_words(G) * sizeof(u64), hipHostMallocDefault));
    memset(r.h_pin, 0, hp_words(G) * sizeof(u64));
    CHKM(ensure(c, r, r.d_off, 2 * OFF_WORDS * sizeof(u64)));
    CHKM(ensure(c, r, r.d_cnt, (G + G * G) * sizeof(u64)));
    CHKM(ensure(c, r, r.d_res, 12 * sizeof(u64)));
    return HJGPU_OK;
}

// RCCL's kernels need CUs WHILE the probe side is being partitioned (the build side arrives then; CPRA: slice i is on
// the links while slice i+1 is partitioned).  K6's workgroups fill a CU's LDS and live until their pass ends, so a
// kernel that arrives in the middle of a pass would wait for its end: the ranks' K6 grids leave some CUs free.
// That is free: K6 runs at the memory system's rate, not the CUs' - 64 M x 1 G on one GPU, ms per pass with
// 0 / 8 / 16 / 32 CUs left out: 3.01 / 3.03 / 3.03 / 3.13 and 3.04 / 3.03 / 3.03 / 3.06 (profiles/r02_reserve_sweep.txt).
int apply_reserve(hjgpu_comm *c)
{
    const bool rccl_ = c->transport && strcmp(c->transport->name(), "rccl") == 0;
    const int n = c->reserve_cus >= 0 ? c->reserve_cus : (rccl_ && c->nranks > 1 ? 16 : 0);
    char v[16];
    snprintf(v, sizeof(v), "%d", n);
    for (Rank &r : c->ranks) {
        if (r.join && hjgpu_set_option(r.join, "reserve_cus", v) != HJGPU_OK) return cfail(c, HJGPU_EINVAL, "reserve_cus");
        if (r.part && hjgpu_set_option(r.part, "reserve_cus", v) != HJGPU_OK) return cfail(c, HJGPU_EINVAL, "reserve_cus");
    }
    return HJGPU_OK;
}

void destroy_rank(Rank &r)
{
    (void)hipSetDevice(r.device);
    (void)hipDeviceSynchronize();
    Buf *bufs[] = {&r.rbuf, &r.send_k[0], &r.send_k[1], &r.send_v[0], &r.send_v[1], &r.recv_k[0], &r.recv_k[1],
                   &r.recv_v[0], &r.recv_v[1], &r.rsend_k, &r.rsend_v, &r.rrecv_k, &r.rrecv_v, &r.d_off, &r.d_cnt,
                   &r.d_res, &r.scratch, &r.pre, &r.shard[0], &r.shard[1], &r.shard[2], &r.shard[3], &r.rows_col[0],
                   &r.rows_col[1], &r.rows_col[2], &r.cnt2[0], &r.cnt2[1], &r.cnt2_all[0], &r.cnt2_all[1], &r.cnt_recv[0], &r.cnt_recv[1]};
    for (Buf *b : bufs) if (b->p) (void)hipFree(b->p);
    hipEvent_t evs[] = {r.ev_ready, r.ev_x0, r.ev_x1, r.ev_t0, r.ev_t1, r.ev_part[0], r.ev_part[1], r.ev_xchg[0],
                        r.ev_xchg[1], r.ev_join[0], r.ev_join[1], r.ev_rx, r.ev_dbg, r.ev_dbg2, r.ev_up_s, r.ev_up_r, r.lb_in, r.lb_out};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r.ev_w) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r.ev_up_slice) if (e) (void)hipEventDestroy(e);
    for (hipStream_t s : {r.main, r.comm, r.prep, r.up}) if (s) (void)hipStreamDestroy(s);
    if (r.h_pin) (void)hipHostFree(r.h_pin);
    if (r.join) (void)hjgpu_destroy(r.join);
    if (r.part) (void)hjgpu_destroy(r.part);
}

// thread_beg / thread_end, npj.cpp:516-529
void range_of(size_t n, size_t alignment, size_t t, size_t T, size_t *beg, size_t *end)
{
    const size_t part = (n / T) & ~(alignment - 1);
    *beg = part * t;
    *end = t + 1 == T ? n : part * (t + 1);
}

std::vector<hipStream_t> streams_of(hjgpu_comm *c, hipStream_t Rank::*which)
{
    std::vector<hipStream_t> s;
    for (Rank &r : c->ranks) s.push_back(r.*which);
    return s;
}

// ---- host-side waits with a deadline ------------------------------------------------------------------------
// The communicator gives up: whatever RCCL has in flight is cancelled (ncclCommAbort), every later call fails fast.
int give_up(hjgpu_comm *c, const char *why)
{
    // the local ranks' host threads (each_rank) may hit the deadline or an asynchronous error at the same time: the
    // abort runs exactly once, the first caller's reason stays
    {
        std::lock_guard<std::mutex> g(c->abort_mu);
        if (!c->broken) {
            snprintf(c->why_broken, sizeof(c->why_broken), "%s", why);
            if (c->transport) c->transport->abort();
            c->broken = true;
        }
    }
    return cfail(c, HJGPU_ERCCL, c->why_broken);
}

int refuse_broken(hjgpu_comm *c)
{
    if (c && c->broken) return cfail(c, HJGPU_ERCCL, "the communicator was aborted", c->why_broken);
    return HJGPU_OK;
}

struct Waited { int local; hipStream_t stream; const char *name; };

// Blocks until every listed stream has drained.  Without a deadline (timeout_ms = 0) that is hipStreamSynchronize;
// with one the streams are polled, RCCL is asked for asynchronous errors, and at the deadline the communicator is
// aborted: the call returns HJGPU_ERCCL naming the rank and stream that did not finish.
int wait_for(hjgpu_comm *c, const std::vector<Waited> &ws)
{
    char text[400];
    if (c->timeout_ms <= 0) {
        for (const Waited &w : ws) {
            HIPM(c, hipSetDevice(c->ranks[(size_t)w.local].device));
            HIPM(c, hipStreamSynchronize(w.stream));
        }
        if (c->transport && c->transport->async_error(text, sizeof(text))) return give_up(c, text);
        return HJGPU_OK;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(c->timeout_ms);
    std::vector<bool> done(ws.size(), false);
    size_t left = ws.size();
    unsigned spins = 0;
    while (left) {
        for (size_t i = 0; i < ws.size(); ++i) {
            if (done[i]) continue;
            HIPM(c, hipSetDevice(c->ranks[(size_t)ws[i].local].device));
            const hipError_t e = hipStreamQuery(ws[i].stream);
            if (e == hipSuccess) { done[i] = true; --left; }
            else if (e != hipErrorNotReady) return cfail(c, HJGPU_EHIP, "hipStreamQuery", hipGetErrorString(e));
            else (void)hipGetLastError();
        }
        if (!left) break;
        if (c->transport && c->transport->async_error(text, sizeof(text))) return give_up(c, text);
        if (std::chrono::steady_clock::now() > deadline) {
            for (size_t i = 0; i < ws.size(); ++i)
                if (!done[i]) {
                    snprintf(text, sizeof(text), "deadline of %d ms expired: the %s stream of rank %d did not finish (transport %s, %d ranks); "
                                                 "communicator aborted", c->timeout_ms, ws[i].name, c->ranks[(size_t)ws[i].local].global,
                             c->transport ? c->transport->name() : "?", c->nranks);
                    break;
                }
            return give_up(c, text);
        }
        // the first polls spin (a step is a few milliseconds), later ones sleep
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(100));
        else std::this_thread::yield();
    }
    return HJGPU_OK;
}

int wait_stream(hjgpu_comm *c, int local, hipStream_t s, const char *name)
{
    return wait_for(c, std::vector<Waited>{{local, s, name}});
}

int sync_all(hjgpu_comm *c)
{
    std::vector<Waited> ws;
    for (int l = 0; l < (int)c->ranks.size(); ++l) {
        Rank &r = c->ranks[(size_t)l];
        ws.push_back({l, r.up, "upload"});
        ws.push_back({l, r.prep, "partitioning"});
        ws.push_back({l, r.comm, "exchange"});
        ws.push_back({l, r.main, "join"});
    }
    return wait_for(c, ws);
}

// A rank's local join whose plan groups (hjgpu_phj_overlapped_async) is planned on the device and enqueue-only; should one of its groups
// have been larger than the plan's workspace (heavy duplicates), the join is done again host-planned - here, by the rank's thread, once
// the stream is known to be idle (a wait with the communicator's deadline: the inputs' arrival is behind it, what follows waits for
// nothing but this device)
int settle_grouped_local_join(hjgpu_comm *c, int l, Rank &r, size_t inner, size_t outer, const hjgpu_phj_params *prm)
{
    uint32_t groups = 0;
    if (hjgpu_grouped_plan(r.join, inner, outer, prm, &groups) != HJGPU_OK || !groups) return HJGPU_OK;
    const int w = wait_stream(c, l, r.main, "join");
    if (w != HJGPU_OK) return w;
    const int st = hjgpu_get_async_status(r.join, r.main);
    // (a zero key / an overflowing result column travel with the flags: hjgpu_accumulate_async_status)
    if (st != HJGPU_OK && st != HJGPU_EOVERFLOW && st != HJGPU_EZEROKEY) return cfail(c, st, "grouped local join", hjgpu_last_error(r.join));
    return HJGPU_OK;
}

void add_stats(hjgpu_stats *acc, const hjgpu_stats &s)
{
    acc->ms_total += s.ms_total; acc->ms_histogram += s.ms_histogram; acc->ms_plan += s.ms_plan;
    acc->ms_scatter1 += s.ms_scatter1; acc->ms_scatter2 += s.ms_scatter2; acc->ms_join += s.ms_join;
    acc->ms_build += s.ms_build; acc->ms_close_gaps += s.ms_close_gaps; acc->ms_inner_wait += s.ms_inner_wait;
    acc->ms_scatter0 += s.ms_scatter0;
    acc->fanout1 = s.fanout1; acc->fanout2 = s.fanout2; acc->buckets = s.buckets; acc->ms_reserve = s.ms_reserve;
    acc->groups = s.groups;                                          // (a rank's local join by a grouped plan)
    acc->placement_tried = s.placement_tried; acc->placement_timeboxed = s.placement_timeboxed;
    acc->placement_fill_ms = s.placement_fill_ms; acc->placement_search_ms = s.placement_search_ms; acc->placement_bytes = s.placement_bytes;
}

float elapsed(hipEvent_t a, hipEvent_t b)
{
    float ms = 0;
    return hipEventElapsedTime(&ms, a, b) == hipSuccess ? ms : 0.f;
}

// Global sum of the ranks' running results and status flags (d_res[0..5]) -> host, on the join streams.  The flags
// travel with the aggregates, so EVERY rank learns that SOME rank met a zero build key (npj.cpp:196-210: such a
// tuple is not in the table) or overflowed its result columns, and every rank returns the same status.
// Before the reduction every rank's own result (its rows, for materialised joins) goes to hp_local().
int reduce_results(hjgpu_comm *c, hjgpu_result *result)
{
    const size_t G = (size_t)c->nranks;
    std::vector<u64 *> res;
    for (Rank &r : c->ranks) {
        res.push_back(static_cast<u64 *>(r.d_res.p));
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hipMemcpyAsync(hp_local(r, G), r.d_res.p, 6 * sizeof(u64), hipMemcpyDeviceToHost, r.main));
    }
    const std::vector<hipStream_t> mains = streams_of(c, &Rank::main);
    CHKM(c->transport->all_reduce_u64(res.data(), 6, mains.data()));
    for (Rank &r : c->ranks) {
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hipMemcpyAsync(hp_result(r, G), r.d_res.p, 6 * sizeof(u64), hipMemcpyDeviceToHost, r.main));
    }
    CHKM(sync_all(c));
    const u64 *h = hp_result(c->ranks[0], G);
    if (result) { result->count = h[0]; result->sum_keys = h[1]; result->sum_outer_vals = h[2]; result->sum_inner_vals = h[3]; }
    if (h[4]) return cfail(c, HJGPU_EZEROKEY, "NPJ: a build key is 0, the empty-bucket sentinel (npj.cpp:583): that tuple is not in the table");
    if (h[5]) return cfail(c, HJGPU_EOVERFLOW, "materialised output exceeded the capacity of some rank's result columns (rows needed per rank: hjgpu_shard_rows.rows)");
    return HJGPU_OK;
}

int check_rows(hjgpu_comm *c, const hjgpu_shard_rows *rows)
{
    if (!rows) return HJGPU_OK;
    for (size_t l = 0; l < c->ranks.size(); ++l) {
        const hjgpu_output &o = rows[l].out;
        if (!o.d_keys || !o.d_outer_vals || !o.d_inner_vals) return cfail(c, HJGPU_EINVAL, "null result column in a rank's hjgpu_shard_rows");
        const size_t bs = o.block_size ? o.block_size : 65536;
        if (bs < 256 || (bs & (bs - 1)) || o.capacity < bs) return cfail(c, HJGPU_EINVAL, "result columns: block_size must be a power of two >= 256 and capacity at least one block");
    }
    return HJGPU_OK;
}

// ---- PHJ / NPJ: replicated build side, sharded probe side ---------------------------------------------
// `inner_ready[l]` (optional): an event after which rank l's columns are valid (hjgpu_join_host_multi's uploads) -
// the probe shard's for every rank, the build columns' as well on the root.
int replicated_join(hjgpu_comm *c, int algorithm, const hjgpu_shard *shards, hjgpu_shard_rows *rows, int root,
                    const hjgpu_phj_params *pp, const hjgpu_npj_params *np, hjgpu_result *result,
                    hjgpu_multi_stats *stats, bool from_host = false)
{
    if (!c || !shards) return HJGPU_EINVAL;
    CHKM(refuse_broken(c));
    if (root < 0 || root >= c->nranks) return cfail(c, HJGPU_EINVAL, "root is not a rank of this communicator");
    CHKM(check_rows(c, rows));
    const auto t0 = std::chrono::steady_clock::now();
    const int L = (int)c->ranks.size();
    const size_t G = (size_t)c->nranks;
    const size_t inner = shards[0].inner;
    for (int l = 0; l < L; ++l) {
        if (shards[l].inner != inner) return cfail(c, HJGPU_EINVAL, "the build side has the same size on every rank");
        if (c->ranks[l].global == root && inner && (!shards[l].d_inner_keys || !shards[l].d_inner_vals))
            return cfail(c, HJGPU_EINVAL, "the root's build columns are missing");
    }
    // keys | payloads in ONE buffer: one exchange (each RCCL kernel has to find free CUs next to the persistent
    // partitioning kernels of the probe side: two get in during the first millisecond of a step, four would not)
    const size_t stride = (inner + 4 + 3) & ~size_t(3);             // payloads stay 16-byte aligned
    const size_t bytes = 2 * stride * sizeof(uint32_t);
    std::vector<void *> bufs;
    for (int l = 0; l < L; ++l) {
        Rank &r = c->ranks[l];
        CHKM(ensure(c, r, r.rbuf, c->transport->replicate_capacity(bytes)));
        bufs.push_back(r.rbuf.p);
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hj_zero_async(r.d_res.p, 12 * sizeof(u64), r.main));
        HIPM(c, hipEventRecord(r.ev_x0, r.comm));
        if (from_host) {
            HIPM(c, hipStreamWaitEvent(r.main, r.ev_up_s, 0));       // the probe shard is in HBM
            if (r.global == root) HIPM(c, hipStreamWaitEvent(r.comm, r.ev_up_r, 0));
        }
        if (r.global == root && inner) {
            uint32_t *b = static_cast<uint32_t *>(r.rbuf.p);
            HIPM(c, hj_copy_async(b, shards[l].d_inner_keys, inner * sizeof(uint32_t), r.comm));
            HIPM(c, hj_copy_async(b + stride, shards[l].d_inner_vals, inner * sizeof(uint32_t), r.comm));
        }
    }
    const std::vector<hipStream_t> comms = streams_of(c, &Rank::comm);
    if (inner) CHKM(c->transport->replicate(bufs.data(), bytes, root, c->ring_broadcast, comms.data()));
    CHKM(each_rank(L, [&](int l) -> int {
        Rank &r = c->ranks[l];
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hipEventRecord(r.ev_x1, r.comm));
        HIPM(c, hipEventRecord(r.ev_ready, r.comm));
        const uint32_t *rk = static_cast<const uint32_t *>(r.rbuf.p), *rv = rk + stride;
        u64 *acc = static_cast<u64 *>(r.d_res.p);
        hjgpu_result *d_res = reinterpret_cast<hjgpu_result *>(acc);
        if (from_host) HIPM(c, hipEventRecord(r.ev_t0, r.main));    // the join starts here: before the upload has ended?
        if (rows) JOINM(c, r.join, hjgpu_set_async_output(r.join, &rows[l].out));
        if (algorithm == 1) {
            // the probe shard is histogrammed and partitioned while the build side is still arriving
            JOINM(c, r.join, hjgpu_phj_overlapped_async(r.join, rk, rv, inner, shards[l].d_outer_keys, shards[l].d_outer_vals,
                                                        shards[l].outer, pp, d_res, r.main, r.ev_ready));
            const int sg = settle_grouped_local_join(c, l, r, inner, shards[l].outer, pp);
            if (sg != HJGPU_OK) return sg;
        } else {
            HIPM(c, hipStreamWaitEvent(r.main, r.ev_ready, 0));     // NPJ builds first: it needs all of R
            JOINM(c, r.join, hjgpu_npj_async(r.join, rk, rv, inner, shards[l].d_outer_keys, shards[l].d_outer_vals,
                                             shards[l].outer, np, d_res, r.main));
        }
        // a zero build key / an overflowing result column on ANY rank fails the call on EVERY rank
        JOINM(c, r.join, hjgpu_accumulate_async_status(r.join, reinterpret_cast<uint64_t *>(acc + 4), r.main));
        return HJGPU_OK;
    }));
    const int status = reduce_results(c, result);
    if (status != HJGPU_OK && status != HJGPU_EZEROKEY && status != HJGPU_EOVERFLOW) return status;
    if (rows) for (int l = 0; l < L; ++l) rows[l].rows = hp_local(c->ranks[l], G)[0];
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        Rank &r = c->ranks[0];
        HIPM(c, hipSetDevice(r.device));
        stats->ms_exchange = elapsed(r.ev_x0, r.ev_x1);
        hjgpu_stats js;
        if (hjgpu_get_stats(r.join, &js) == HJGPU_OK) { add_stats(&stats->join, js); stats->ms_exchange_wait = js.ms_inner_wait; }
        stats->joins = 1;
        stats->tuples_joined = inner + shards[0].outer;
        // scatter + all-gather: the root sends G-1 slices, then every rank its own slice to G-1 peers; ring: one copy on
        const u64 per = c->transport->replicate_capacity(bytes) / c->nranks;
        if (c->nranks > 1 && inner)
            stats->bytes_sent = c->ring_broadcast ? (u64)bytes : (u64)(r.global == root ? 2 : 1) * (c->nranks - 1) * per;
        stats->ms_wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return status;
}

// ---- CPRA: both sides chunked, co-partitioned by an all-to-all-v -------------------------------------------
// One relation slice per local rank on its way through the exchange.
struct Slice {
    const uint32_t *keys, *vals;     // the rank's input slice
    size_t n;
};

// which = 0: the build side's buffers; 1 + slot: the probe side's double-buffered slots
struct ExchangeBufs { Buf *sk, *sv, *rk, *rv; hipEvent_t done; };
ExchangeBufs bufs_of(Rank &r, int which)
{
    if (which == 0) return {&r.rsend_k, &r.rsend_v, &r.rrecv_k, &r.rrecv_v, r.ev_rx};
    const int slot = which - 1;
    return {&r.send_k[slot], &r.send_v[slot], &r.recv_k[slot], &r.recv_v[slot], r.ev_xchg[slot]};
}

struct CpraStep {
    hjgpu_comm *c;
    int L, G;
    std::vector<std::vector<u64>> soff, scnt, roff, rcnt;    // [local rank][peer]
    std::vector<u64> recv_total;
    hjgpu_multi_stats *stats;
    // the pieces of what local rank l received in the LAST exchange: piece p (from rank p) = rows [pieces[l][p], pieces[l][p + 1])
    std::vector<std::vector<u64>> pieces;
    // ... of the array `base[l]`, the first of them at row pieces[l][0].  One-level plan, exchange in place: the array is the
    // rank's SEND buffer - its own partitions were written last (hjgpu_partition_packed_own_last_async), the other ranks'
    // pieces are received right behind them: piece 0 = the own one, never copied; pieces 1.. = the other ranks in order.
    std::vector<const void *> base;
    bool exchange_in_flight = false;                          // local rank 0's ev_x0 / ev_x1 hold an unread exchange
    bool from_host = false;                                   // the inputs are being uploaded (hjgpu_join_host_multi)
    float exchange_ms = 0;
    // One-level plan (1 <= G <= 8): the exchange-level partitioning has fan-out G * k and packed output - it IS pass 1 of
    // the join (the reference's own-chunk partitioning is the join's partitioning, cpra2.cpp:1757-1827; ownership
    // 1868-1872): rank g owns partitions [g k, (g + 1) k), the receiver runs pass 2 + build / probe on the pieces as
    // they arrived (hjgpu_phj_build_prepartitioned).  k = 0: round 2's two-level plan (fan-out G, complete local PHJ).
    uint32_t k = 0;
    uint32_t fanout() const { return k ? (uint32_t)G * k : (uint32_t)G; }
    // Counts published with the partitions (cpra2.cpp:1783-1840): when the whole fused histogram G * k * F2 fits K4's LDS
    // (<= 32768 counters; F2 = what the largest receiver's build side needs, the same on every rank) the SENDER's
    // histogram pass counts the receivers' final partitions in the same read of its keys, the histograms travel with the
    // counts all-gather, and a receiver joins what arrived without a histogram pass of its own (K4p: 8 of the 52 bytes
    // per tuple at world 1).  Probe slices only (the build side's K4p reads 1 / 16 of the bytes).
    bool fused = false;
    uint32_t F2 = 0, factor2 = 0;
    std::vector<char> counts_usable = std::vector<char>(2, 0);   // [slot] the last exchange through the slot delivered counts
    CpraStep(hjgpu_comm *comm, hjgpu_multi_stats *st)
        : c(comm), L((int)comm->ranks.size()), G(comm->nranks), soff(L, std::vector<u64>(G)), scnt(L, std::vector<u64>(G)),
          roff(L, std::vector<u64>(G)), rcnt(L, std::vector<u64>(G)), recv_total(L), stats(st), pieces(L, std::vector<u64>(G + 1)),
          base(L, nullptr) {}

    // call when local rank 0's exchange stream is known to be idle
    void note_exchange()
    {
        if (!exchange_in_flight) return;
        Rank &r = c->ranks[0];
        (void)hipSetDevice(r.device);
        exchange_ms += elapsed(r.ev_x0, r.ev_x1);
        exchange_in_flight = false;
    }

    // local partition with fan-out G (cpra2.cpp:1757-1827 on the rank's own chunk) -> counts to every rank
    // (cpra2.cpp:1834-1840) -> all-to-all-v of keys and payloads (the gather of cpra2.cpp:1891-1904 / 1946-1959).
    // `slot`: which offsets / events; the call returns with the transfers enqueued.
    // `ready`: which upload event of the rank the partitioning waits for (host path), or nullptr.
    int exchange(const std::vector<Slice> &in, int which, int slot, hipEvent_t Rank::*ready = nullptr)
    {
        CHKM(begin_exchange(in, which, slot, ready, -1));
        return finish_exchange(in, which, slot);
    }

    // first half: the partitioning of every local rank's slice is enqueued (nothing waits on the host)
    // `host_slice` >= 0 (host path): the upload event of that probe slice instead (Rank::ev_up_slice)
    int begin_exchange(const std::vector<Slice> &in, int which, int slot, hipEvent_t Rank::*ready = nullptr, int host_slice = -1)
    {
        const size_t Gs = (size_t)G;
        const size_t F = fanout();                                  // partitions of the exchange-level pass
        const size_t tuple_bytes = k ? sizeof(u64) : sizeof(uint32_t);
        const bool own_last = k && c->exchange_in_place && !c->self_via_rccl;
        return each_rank(L, [&](int l) -> int {
            Rank &r = c->ranks[l];
            ExchangeBufs b = bufs_of(r, which);
            size_t hold = in[l].n;
            if (own_last) {
                // room for what the others will send, behind the rank's own output: as much as the last exchange through
                // this buffer needed, or - first time - the others' shares of a chunk like this one, with an eighth of headroom
                const u64 guess = (u64)in[l].n + (u64)in[l].n / Gs * (Gs - 1) * 9 / 8 + 4096;
                hold = (size_t)(r.want_rows[which] > guess ? r.want_rows[which] : guess);
            }
            CHKM(ensure_scatter_target(c, r, *b.sk, (hold + 16) * tuple_bytes));
            if (!k) CHKM(ensure(c, r, *b.sv, (in[l].n + 4) * sizeof(uint32_t)));
            u64 *d_off = static_cast<u64 *>(r.d_off.p) + (size_t)slot * OFF_WORDS;
            u64 *h_off = hp_off(r, Gs, slot);
            HIPM(c, hipSetDevice(r.device));
            if (host_slice >= 0) HIPM(c, hipStreamWaitEvent(r.prep, r.ev_up_slice[(size_t)host_slice], 0));
            else if (ready) HIPM(c, hipStreamWaitEvent(r.prep, r.*ready, 0));
            if (c->debug_serialize & 6) {
                HIPM(c, hipEventRecord(r.ev_dbg, r.main));
                if (c->debug_serialize & 2) HIPM(c, hipStreamWaitEvent(r.prep, r.ev_dbg, 0));
                if (c->debug_serialize & 4) HIPM(c, hipStreamWaitEvent(r.comm, r.ev_dbg, 0));
            }
            // in place: the slot's previous slice has been joined (its own piece lives in this buffer)
            if (own_last && which) HIPM(c, hipStreamWaitEvent(r.prep, r.ev_join[slot], 0));
            const bool counted = fused && which;
            if (counted) {
                CHKM(ensure(c, r, r.cnt2[slot], F * F2 * sizeof(u64)));
                CHKM(ensure(c, r, r.cnt2_all[slot], Gs * F * F2 * sizeof(u64)));
                CHKM(ensure(c, r, r.cnt_recv[slot], Gs * k * F2 * sizeof(u64)));
            }
            if (in[l].n && counted)
                JOINM(c, r.part, hjgpu_partition_packed_counted_async(r.part, in[l].keys, in[l].vals, in[l].n, TOP_LEVEL_FACTOR, (uint32_t)F,
                                                                      own_last ? (uint32_t)r.global * k : 0u, own_last ? k : 0u, factor2, F2,
                                                                      static_cast<uint64_t *>(b.sk->p), reinterpret_cast<uint64_t *>(d_off),
                                                                      static_cast<uint64_t *>(r.cnt2[slot].p), r.prep));
            else if (in[l].n && own_last)
                JOINM(c, r.part, hjgpu_partition_packed_own_last_async(r.part, in[l].keys, in[l].vals, in[l].n, TOP_LEVEL_FACTOR, (uint32_t)F,
                                                                       (uint32_t)r.global * k, k, static_cast<uint64_t *>(b.sk->p),
                                                                       reinterpret_cast<uint64_t *>(d_off), r.prep));
            else if (in[l].n && k)
                JOINM(c, r.part, hjgpu_partition_packed_async(r.part, in[l].keys, in[l].vals, in[l].n, TOP_LEVEL_FACTOR, (uint32_t)F,
                                                              static_cast<uint64_t *>(b.sk->p), reinterpret_cast<uint64_t *>(d_off), r.prep));
            else if (in[l].n)
                JOINM(c, r.part, hjgpu_partition_async(r.part, in[l].keys, in[l].vals, in[l].n, TOP_LEVEL_FACTOR, (uint32_t)G,
                                                       static_cast<uint32_t *>(b.sk->p), static_cast<uint32_t *>(b.sv->p),
                                                       reinterpret_cast<uint64_t *>(d_off), r.prep));
            else {
                HIPM(c, hj_zero_async(d_off, (F + 1) * sizeof(u64), r.prep));
                if (counted) HIPM(c, hj_zero_async(r.cnt2[slot].p, F * F2 * sizeof(u64), r.prep));
            }
            HIPM(c, hipMemcpyAsync(h_off, d_off, (F + 1) * sizeof(u64), hipMemcpyDeviceToHost, r.prep));
            HIPM(c, hipEventRecord(r.ev_part[slot], r.prep));
            return HJGPU_OK;
        });
    }

    // second half: the host waits for the counts (how much every peer gets decides the receive buffers), the ranks
    // exchange them, the transfers are enqueued
    int finish_exchange(const std::vector<Slice> &in, int which, int slot)
    {
        const size_t Gs = (size_t)G;
        const size_t tuple_bytes = k ? sizeof(u64) : sizeof(uint32_t);
        const bool own_last = k && c->exchange_in_place && !c->self_via_rccl;
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            CHKM(wait_stream(c, l, r.prep, "partitioning"));
            const u64 *h_off = hp_off(r, Gs, slot);
            hj_exchange::send_layout(h_off, k, G, r.global, (u64)in[l].n, own_last, soff[l].data(), scnt[l].data());
            if (l == 0 && stats) {
                hjgpu_stats ps;
                if (in[l].n && hjgpu_get_stats(r.part, &ps) == HJGPU_OK) stats->ms_partition += ps.ms_total;
                for (int p = 0; p < G; ++p) if (p != r.global) stats->bytes_sent += scnt[l][p] * 8;
            }
        }
        // counts first, payload second: one small all-gather gives every rank the G x G matrix of message sizes
        std::vector<const void *> csend;
        std::vector<void *> crecv;
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            u64 *d_cnt = static_cast<u64 *>(r.d_cnt.p);
            HIPM(c, hipSetDevice(r.device));
            u64 *h_cnt = hp_cnt(r, Gs);                             // pinned: the copy is a DMA that runs later
            memcpy(h_cnt, scnt[l].data(), Gs * sizeof(u64));
            HIPM(c, hipMemcpyAsync(d_cnt, h_cnt, Gs * sizeof(u64), hipMemcpyHostToDevice, r.comm));
            csend.push_back(d_cnt); crecv.push_back(d_cnt + Gs);
        }
        const std::vector<hipStream_t> comms = streams_of(c, &Rank::comm);
        CHKM(c->transport->all_gather(csend.data(), crecv.data(), Gs * sizeof(u64), comms.data()));
        const bool counted = fused && which;
        if (counted) {
            // every rank's fused histogram of its slice to every rank (the partitioning streams have been waited for above)
            std::vector<const void *> hs;
            std::vector<void *> hr;
            for (int l = 0; l < L; ++l) { hs.push_back(c->ranks[l].cnt2[slot].p); hr.push_back(c->ranks[l].cnt2_all[slot].p); }
            CHKM(c->transport->all_gather(hs.data(), hr.data(), (size_t)fanout() * F2 * sizeof(u64), comms.data()));
        }
        if (which) counts_usable[slot] = counted;
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            HIPM(c, hipSetDevice(r.device));
            HIPM(c, hipMemcpyAsync(hp_matrix(r, Gs), static_cast<u64 *>(r.d_cnt.p) + Gs, Gs * Gs * sizeof(u64),
                                   hipMemcpyDeviceToHost, r.comm));
        }
        std::vector<const void *> ks, vs;
        std::vector<void *> kr, vr;
        std::vector<const u64 *> so, sc, ro, rc;
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            ExchangeBufs b = bufs_of(r, which);
            CHKM(wait_stream(c, l, r.comm, "exchange"));            // also: the previous exchange has left the links
            HIPM(c, hipSetDevice(r.device));
            if (l == 0) note_exchange();
            const u64 *matrix = hp_matrix(r, Gs);                   // matrix[src][dst]
            // (the transport is told nothing about a message that stays: in place, counts for itself are 0 on both sides)
            const hj_exchange::Receive rx = hj_exchange::receive_layout(matrix, G, r.global, (u64)in[l].n, own_last,
                                                                        b.sk->cap / tuple_bytes > 16 ? b.sk->cap / tuple_bytes - 16 : 0,
                                                                        roff[l].data(), rcnt[l].data(), pieces[l].data());
            const u64 at = rx.rows;
            recv_total[l] = at;
            // too small this time (the others sent more than this rank's own chunk suggested): the copying path, and the
            // buffer grows before the next exchange through it
            if (own_last) r.want_rows[which] = rx.in_place ? (r.want_rows[which] > rx.need ? r.want_rows[which] : rx.need) : rx.need + rx.need / 4;
            if (rx.in_place) {
                scnt[l][r.global] = 0;
                base[l] = b.sk->p;
                ks.push_back(b.sk->p); vs.push_back(b.sv->p); kr.push_back(b.sk->p); vr.push_back(b.rv->p);
            } else {
                // a quarter of headroom: the next slices rarely need a new allocation (hipFree waits for the device)
                if ((at + 16) * tuple_bytes > b.rk->cap) CHKM(ensure(c, r, *b.rk, (at + at / 4 + 16) * tuple_bytes));
                if (!k && (at + 4) * sizeof(uint32_t) > b.rv->cap) CHKM(ensure(c, r, *b.rv, (at + at / 4 + 4) * sizeof(uint32_t)));
                base[l] = b.rk->p;
                if (l == 0 && stats && rcnt[l][r.global]) stats->self_copies += 1;
                ks.push_back(b.sk->p); vs.push_back(b.sv->p); kr.push_back(b.rk->p); vr.push_back(b.rv->p);
            }
            so.push_back(soff[l].data()); sc.push_back(scnt[l].data()); ro.push_back(roff[l].data()); rc.push_back(rcnt[l].data());
            HIPM(c, hipStreamWaitEvent(r.comm, r.ev_part[slot], 0));
            if (which) HIPM(c, hipStreamWaitEvent(r.comm, r.ev_join[slot], 0));   // the slot's previous slice has been joined
            if (counted) {
                // the rows of the senders' histograms that describe MY partitions, in the order of my pieces (in place: the
                // own piece first, then the other ranks in order; else rank order): what K4p would have counted
                const size_t mine = (size_t)k * F2, all = (size_t)fanout() * F2;
                int piece = 0;
                auto take = [&](int sender) -> int {
                    HIPM(c, hj_copy_async(static_cast<u64 *>(r.cnt_recv[slot].p) + (size_t)piece * mine,
                                           static_cast<const u64 *>(r.cnt2_all[slot].p) + (size_t)sender * all + (size_t)r.global * mine,
                                           mine * sizeof(u64), r.comm));
                    ++piece;
                    return HJGPU_OK;
                };
                if (rx.in_place) {
                    CHKM(take(r.global));
                    for (int p = 0; p < G; ++p) if (p != r.global) CHKM(take(p));
                } else
                    for (int p = 0; p < G; ++p) CHKM(take(p));
            }
            if (l == 0) HIPM(c, hipEventRecord(r.ev_x0, r.comm));
        }
        // packed tuples: keys and payloads travel together, ONE all-to-all-v per slice
        CHKM(c->transport->all_to_all_v(ks.data(), so.data(), sc.data(), kr.data(), ro.data(), rc.data(), tuple_bytes, comms.data()));
        if (!k) CHKM(c->transport->all_to_all_v(vs.data(), so.data(), sc.data(), vr.data(), ro.data(), rc.data(), sizeof(uint32_t), comms.data()));
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            HIPM(c, hipSetDevice(r.device));
            if (l == 0) { HIPM(c, hipEventRecord(r.ev_x1, r.comm)); exchange_in_flight = true; }
            HIPM(c, hipEventRecord(bufs_of(r, which).done, r.comm));
        }
        return HJGPU_OK;
    }
};

// from_host: the shards are being uploaded (hjgpu_join_host_multi): the build side's partitioning waits for ev_up_r, the probe
// slices' for ev_up_s - or, with slice_events, slice i for its own upload event ev_up_slice[i] (the shard then arrives in the
// same `slices` pieces, build side first: slice i is partitioned, exchanged and joined while the later slices are on the bus)
int cpra_join(hjgpu_comm *c, const hjgpu_shard *shards, hjgpu_shard_rows *rows, const hjgpu_phj_params *prm, int slices,
              hjgpu_result *result, hjgpu_multi_stats *stats, bool from_host = false, bool slice_events = false)
{
    if (!c || !shards) return HJGPU_EINVAL;
    CHKM(refuse_broken(c));
    // slices exist to overlap the exchange with the local passes; a world of one has nothing to overlap and every slice
    // costs ~0.6 ms of launches, ramps and tails (world 1, 64 M x 1 G: 10.6 / 11.2 / 12.6 / 15.0 ms with 1 / 2 / 4 / 8 slices)
    if (slices <= 0) slices = c->nranks == 1 ? 1 : 4;
    if (slices > 4096) return cfail(c, HJGPU_EINVAL, "at most 4096 slices");
    CHKM(check_rows(c, rows));
    const auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    const int L = (int)c->ranks.size();
    const size_t G = (size_t)c->nranks;
    for (int l = 0; l < L; ++l) {
        const hjgpu_shard &s = shards[l];
        if ((s.inner && (!s.d_inner_keys || !s.d_inner_vals)) || (s.outer && (!s.d_outer_keys || !s.d_outer_vals)))
            return cfail(c, HJGPU_EINVAL, "null column in a shard");
    }
    CpraStep step(c, stats);
    // option "debug_forensics": where this step's audit records start in every rank's two contexts
    std::vector<uint64_t> seq_part((size_t)L, 0), seq_join((size_t)L, 0);
    if (c->debug_forensics)
        for (int l = 0; l < L; ++l) {
            (void)hjgpu_audit_read(c->ranks[l].part, &seq_part[(size_t)l], 0, 0, nullptr, nullptr);
            (void)hjgpu_audit_read(c->ranks[l].join, &seq_join[(size_t)l], 0, 0, nullptr, nullptr);
        }
    // A rank's share beyond two passes' reach (a build side of ~228 M rows per rank and more, with a probe side large enough for a
    // third pass to pay: hjgpu_grouped_plan, the rule of hjgpu_phj) is joined by a GROUPED plan.  Every rank has to take the same
    // road - it decides the shape of the exchange - so the rule is fed what every rank can know: the relations' TOTAL sizes (one
    // all-reduce of two words), a rank's share taken as 1 / G of them.  The grouped road: exchange with fan-out G in separate
    // columns (round 2's two-level plan), the probe side in ONE slice, then the rank's complete local join - whose plan groups.
    bool grouped = false;
    const int asked_slices = slices;
    if (c->cpra_grouped) {
        std::vector<u64 *> words;
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            HIPM(c, hipSetDevice(r.device));
            u64 *h = hp_result(r, G);
            h[0] = shards[l].inner; h[1] = shards[l].outer;
            HIPM(c, hipMemcpyAsync(r.d_cnt.p, h, 2 * sizeof(u64), hipMemcpyHostToDevice, r.comm));
            words.push_back(static_cast<u64 *>(r.d_cnt.p));
        }
        const std::vector<hipStream_t> comms = streams_of(c, &Rank::comm);
        CHKM(c->transport->all_reduce_u64(words.data(), 2, comms.data()));
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            HIPM(c, hipSetDevice(r.device));
            HIPM(c, hipMemcpyAsync(hp_result(r, G), r.d_cnt.p, 2 * sizeof(u64), hipMemcpyDeviceToHost, r.comm));
        }
        for (int l = 0; l < L; ++l) CHKM(wait_stream(c, l, c->ranks[l].comm, "sizes"));
        const u64 *tot = hp_result(c->ranks[0], G);
        uint32_t groups = 0;
        JOINM(c, c->ranks[0].join, hjgpu_grouped_plan(c->ranks[0].join, (size_t)(tot[0] / G), (size_t)(tot[1] / G), prm, &groups));
        grouped = groups > 1;
        // (1: only where the road's extra pass pays - exchange_layout.hpp grouped_road_pays says why and where)
        if (grouped && c->cpra_grouped == 1 && !hj_exchange::grouped_road_pays(tot[0] / G, tot[1] / G, HJGPU_MAX_PARTS)) grouped = false;
        if (grouped) slices = 1;
    }
    // one-level plan while the receiver can take one piece per source rank (<= 8 pieces): fan-out G * k with G * k <= 192,
    // the widest pass 1 whose whole-line carry still fits beside a 16 K-tuple tile
    if (c->nranks <= 8 && !c->cpra_two_level && !grouped) step.k = (uint32_t)(c->cpra_k > 0 && c->cpra_k * c->nranks <= 192 ? c->cpra_k : 192 / c->nranks);
    const uint32_t K = step.k;
    // the layout of what local rank l received (pieces = one per source rank), rows [lo, hi) of it
    auto layout_of = [&](int l, const std::vector<u64> &pieces, u64 lo, u64 hi) {
        hjgpu_prepartitioned lay;
        memset(&lay, 0, sizeof(lay));
        lay.factor1 = TOP_LEVEL_FACTOR; lay.fanout1_total = (uint32_t)c->nranks * K; lay.fanout1 = K;
        lay.first_partition = (uint32_t)c->ranks[l].global * K; lay.chunks = (uint32_t)c->nranks;
        for (int p = 0; p <= c->nranks; ++p) { const u64 x = pieces[(size_t)p]; lay.chunk_offsets[p] = x < lo ? lo : (x > hi ? hi : x); }
        for (int p = c->nranks + 1; p < 9; ++p) lay.chunk_offsets[p] = lay.chunk_offsets[c->nranks];
        return lay;
    };
    for (Rank &r : c->ranks) {
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hj_zero_async(r.d_res.p, 12 * sizeof(u64), r.main));
        // timing events of the slices' waits, one pair per slice, read after the step (nobody waits in between)
        while (r.ev_w.size() < 2 * (size_t)slices) {
            hipEvent_t e = nullptr;
            HIPM(c, hipEventCreate(&e));
            r.ev_w.push_back(e);
        }
    }
    // ---- build side: partition the own chunk -> exchange -> prepared once for all probe slices ---------
    std::vector<Slice> in(L);
    for (int l = 0; l < L; ++l) in[l] = {shards[l].d_inner_keys, shards[l].d_inner_vals, shards[l].inner};
    CHKM(step.exchange(in, 0, 0, from_host ? &Rank::ev_up_r : nullptr));
    const std::vector<u64> inner_recv = step.recv_total;
    const std::vector<std::vector<u64>> inner_pieces = step.pieces;
    const std::vector<const void *> inner_base = step.base;
    // counts published with the partitions (CpraStep::fused): every rank reads the same G x G matrix of the build exchange,
    // so every rank arrives at the same F2 - what the receiver with the largest build side needs
    hjgpu_phj_params prm_fused;
    if (K && c->cpra_fused_counts) {
        const u64 *mx = hp_matrix(c->ranks[0], G);
        u64 most = 0;
        for (size_t dst = 0; dst < G; ++dst) {
            u64 rows_of_dst = 0;
            for (size_t src = 0; src < G; ++src) rows_of_dst += mx[src * G + dst];
            most = most > rows_of_dst ? most : rows_of_dst;
        }
        uint32_t f2 = 0, m2 = 0;
        if (most && hjgpu_prepartitioned_plan(c->ranks[0].join, (size_t)most, K, prm, &f2, &m2) == HJGPU_OK &&
            (u64)G * K * f2 <= 32768 && m2 != TOP_LEVEL_FACTOR) {
            step.fused = true; step.F2 = f2; step.factor2 = m2;
            memset(&prm_fused, 0, sizeof(prm_fused));
            if (prm) prm_fused = *prm;
            prm_fused.fanout2 = f2;                         // every receiver plans the same second level
            prm = &prm_fused;
        }
    }
    std::vector<size_t> max_outer(L);
    for (int l = 0; l < L; ++l) {
        Rank &r = c->ranks[l];
        // batches are the slices this rank RECEIVES: about one local slice when the hash spreads the keys evenly;
        // the workspace is sized for 1.5 of that, larger batches are probed in pieces
        const size_t per = shards[l].outer / (size_t)slices + 16;
        max_outer[l] = (per * 3 / 2 > ((size_t)1 << 20) ? per * 3 / 2 : ((size_t)1 << 20)) & ~size_t(15);
        if (grouped) max_outer[l] = ~size_t(0) >> 1;         // the local join takes what arrived in one call
        HIPM(c, hipSetDevice(r.device));
        // (the grouped road enqueues nothing here: its one local join waits for the probe side's exchange, which the exchange stream
        // runs after the build side's)
        if (!grouped) HIPM(c, hipStreamWaitEvent(r.main, r.ev_rx, 0));
        if (from_host && l == 0) HIPM(c, hipEventRecord(r.ev_t0, r.main));
        if (grouped) continue;                               // no prepared build side: the local join is one whole hjgpu_phj
        if (inner_recv[l] && K) {
            const hjgpu_prepartitioned lay = layout_of(l, inner_pieces[l], inner_pieces[l][0], inner_pieces[l][0] + inner_recv[l]);
            JOINM(c, r.join, hjgpu_phj_build_prepartitioned(r.join, static_cast<const uint64_t *>(inner_base[l]), &lay, max_outer[l], prm, r.main));
        } else if (inner_recv[l])
            JOINM(c, r.join, hjgpu_phj_build(r.join, static_cast<const uint32_t *>(r.rrecv_k.p), static_cast<const uint32_t *>(r.rrecv_v.p),
                                             (size_t)inner_recv[l], max_outer[l], prm, r.main));
    }
    // Phase times of local rank 0's joins: the BUILD and the LAST probe batch are measured (stats->joins /
    // tuples_joined describe exactly those).  The context's events are re-recorded by every call, so reading a
    // slice's times means waiting for that slice - which would hold the single driving host thread until join(i-1)
    // has finished before partition(i+1) can be enqueued, i.e. the pipeline would be measured out of shape.
    // The build's times are read when the host blocks anyway (the first probe slice's partition counts).
    bool build_stats_pending = stats && inner_recv[0] && !grouped;
    // ---- probe side in slices: partition(i+1) | exchange(i) | join(i-1) ----------------------------------
    // R join S = union_i (R join S_i): the slice results add up (add_result_kernel).
    // Materialised rows: a slice's rows follow the rows of the slices before it in the rank's result columns; where
    // they start is known on the host once the previous join has finished (its count), which is long before this
    // slice has arrived.
    std::vector<u64> used(L, 0);                 // rows in the rank's result columns so far
    std::vector<char> row_pending(L, 0);         // hp_local()[0] will hold the rank's running count (char: the ranks' threads write their own element)
    u64 measured = 0;
    auto join_slice = [&](int i, int slot, const std::vector<u64> &got, const std::vector<std::vector<u64>> &got_pieces,
                          const std::vector<const void *> &got_base) -> int {
        measured = 0;
        CHKM(each_rank(L, [&](int l) -> int {
            Rank &r = c->ranks[l];
            HIPM(c, hipSetDevice(r.device));
            if (c->debug_serialize & 8) {            // the join waits for whatever the partitioning stream has been given so far
                HIPM(c, hipEventRecord(r.ev_dbg2, r.prep));
                HIPM(c, hipStreamWaitEvent(r.main, r.ev_dbg2, 0));
            }
            HIPM(c, hipEventRecord(r.ev_w[2 * (size_t)i], r.main));
            HIPM(c, hipStreamWaitEvent(r.main, r.ev_xchg[slot], 0));
            HIPM(c, hipEventRecord(r.ev_w[2 * (size_t)i + 1], r.main));
            const uint32_t *sk = static_cast<const uint32_t *>(r.recv_k[slot].p), *sv = static_cast<const uint32_t *>(r.recv_v[slot].p);
            u64 *acc = static_cast<u64 *>(r.d_res.p);
            if (inner_recv[l])
                for (u64 b = 0; b < got[l]; b += max_outer[l]) {
                    const size_t m = got[l] - b < max_outer[l] ? (size_t)(got[l] - b) : max_outer[l];
                    if (rows) {
                        if (row_pending[l]) {                       // the rows so far: the previous batch's running count
                            CHKM(wait_stream(c, l, r.main, "join"));
                            used[l] = hp_local(r, G)[0];
                            row_pending[l] = false;
                        }
                        const hjgpu_output &o = rows[l].out;
                        const size_t bs = o.block_size ? o.block_size : 65536;
                        hjgpu_output piece = o;
                        piece.block_size = bs;
                        if (used[l] + bs <= o.capacity) {
                            piece.d_keys = o.d_keys + used[l]; piece.d_outer_vals = o.d_outer_vals + used[l]; piece.d_inner_vals = o.d_inner_vals + used[l];
                            piece.capacity = (o.capacity - used[l]) / bs * bs;
                            JOINM(c, r.join, hjgpu_set_async_output(r.join, &piece));
                        } else {
                            // no block left in this rank's columns: the batch is joined WITHOUT output (its count still joins
                            // the running count: hjgpu_shard_rows.rows reports what the rank needs) and the overflow flag is
                            // raised here.  (Pointing the batch at the last block relied on the emitter to flag the overflow,
                            // which it does only when a second block is claimed: a batch with few matches overwrote valid rows
                            // of earlier slices and the call returned HJGPU_OK.)
                            hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, r.main, acc + 5);
                            HIPM(c, hipGetLastError());
                        }
                    }
                    if (K) {
                        // a batch = rows [b, b + m) of what arrived: a contiguous piece of the pieces (still sorted by partition)
                        const u64 first = got_pieces[l][0];
                        const hjgpu_prepartitioned lay = layout_of(l, got_pieces[l], first + b, first + b + m);
                        // the senders' counts describe WHOLE pieces: a slice that is joined in one batch needs no K4p
                        if (step.fused && step.counts_usable[slot] && b == 0 && m == got[l])
                            JOINM(c, r.join, hjgpu_phj_probe_prepartitioned_counted_async(r.join, static_cast<const uint64_t *>(got_base[l]), &lay,
                                                                                          static_cast<const uint64_t *>(r.cnt_recv[slot].p),
                                                                                          reinterpret_cast<hjgpu_result *>(acc + 8), r.main));
                        else
                        JOINM(c, r.join, hjgpu_phj_probe_prepartitioned_async(r.join, static_cast<const uint64_t *>(got_base[l]), &lay,
                                                                              reinterpret_cast<hjgpu_result *>(acc + 8), r.main));
                    } else if (grouped)
                        // the rank's whole local join; its plan groups where the share needs it (the build side has arrived: the exchange
                        // stream received it before the probe side, whose arrival this stream has just waited for)
                    {
                        JOINM(c, r.join, hjgpu_phj_overlapped_async(r.join, static_cast<const uint32_t *>(r.rrecv_k.p), static_cast<const uint32_t *>(r.rrecv_v.p),
                                                                    (size_t)inner_recv[l], sk + b, sv + b, m, prm,
                                                                    reinterpret_cast<hjgpu_result *>(acc + 8), r.main, nullptr));
                        const int sg = settle_grouped_local_join(c, l, r, (size_t)inner_recv[l], m, prm);
                        if (sg != HJGPU_OK) return sg;
                    }
                    else
                    JOINM(c, r.join, hjgpu_phj_probe_async(r.join, sk + b, sv + b, m, reinterpret_cast<hjgpu_result *>(acc + 8), r.main));
                    hipLaunchKernelGGL(add_result_kernel, dim3(1), dim3(64), 0, r.main, acc, acc + 8);
                    HIPM(c, hipGetLastError());
                    if (rows) {
                        JOINM(c, r.join, hjgpu_accumulate_async_status(r.join, reinterpret_cast<uint64_t *>(acc + 4), r.main));
                        HIPM(c, hipMemcpyAsync(hp_local(r, G), acc, sizeof(u64), hipMemcpyDeviceToHost, r.main));
                        row_pending[l] = true;
                    }
                    if (l == 0) measured = m;                        // the context's events describe its LAST batch
                }
            HIPM(c, hipEventRecord(r.ev_join[slot], r.main));
            return HJGPU_OK;
        }));
        if (c->debug_serialize & 1) for (int l = 0; l < L; ++l) CHKM(wait_stream(c, l, c->ranks[l].main, "join"));
        return HJGPU_OK;
    };
    // option "debug_forensics" = 2 (see hjgpu_comm::frozen): the last record of a context, read behind `stream`; when a partition check of it
    // differs from what the call read - or a probe's result from the batch it read - the checks are done again at once, device quiet
    auto freeze_check = [&](int l, int which, int slice) -> int {
        Rank &r = c->ranks[l];
        hjgpu_ctx *ctx = which ? r.join : r.part;
        HIPM(c, hipSetDevice(r.device));
        uint64_t next = 0;
        JOINM(c, ctx, hjgpu_audit_read(ctx, &next, 0, 0, nullptr, nullptr));
        if (next <= (which ? seq_join : seq_part)[(size_t)l]) return HJGPU_OK;          // no call of this step yet
        uint64_t rec[32];
        JOINM(c, ctx, hjgpu_audit_read(ctx, nullptr, next - 1, 1, rec, which ? r.main : r.prep));
        auto differs = [&](int stage, int input) { return rec[4 * stage] != 0 || rec[4 * stage + 1] != rec[4 * input + 1] || rec[4 * stage + 2] != rec[4 * input + 2] || rec[4 * stage + 3] != rec[4 * input + 3]; };
        const uint64_t kind = rec[4 * 7 + 1];
        bool bad = false;
        if (kind == 3) bad = differs(1, 0);                                              // a partitioning call: output vs input
        if (kind == 1) bad = differs(5, 3);                                              // the build: final partitions vs what it read
        // a probe: final partitions vs the batch, and the result vs the batch (the stress workload: unique build keys, selectivity 1)
        if (kind == 2) {
            bad = differs(2, 0) || rec[4 * 6] != rec[3] || rec[4 * 6 + 1] != rec[1] || rec[4 * 6 + 2] != rec[2];
            // ... and the prepared build side as this probe found it vs what its build read (the step's first join call; nobody waited
            // for the build when it was enqueued: its arrival overlaps the first slice's partitioning)
            uint64_t b[32];
            JOINM(c, ctx, hjgpu_audit_read(ctx, nullptr, seq_join[(size_t)l], 1, b, r.main));
            if (b[4 * 7 + 1] == 1)
                for (int w = 0; w < 4; ++w) if (rec[4 * 5 + w] != (w ? b[4 * 3 + w] : 0)) bad = true;
        }
        if (!bad) return HJGPU_OK;
        size_t n = 0;
        JOINM(c, ctx, hjgpu_audit_recheck(ctx, nullptr, 0, &n));
        const size_t at = c->frozen.size();
        c->frozen.resize(at + 4 + 32 + 9 * n, 0);
        c->frozen[at] = (u64)r.global; c->frozen[at + 1] = (u64)which; c->frozen[at + 2] = (u64)slice; c->frozen[at + 3] = (u64)n;
        memcpy(c->frozen.data() + at + 4, rec, sizeof(rec));
        if (n) JOINM(c, ctx, hjgpu_audit_recheck(ctx, reinterpret_cast<uint64_t *>(c->frozen.data() + at + 36), n, &n));
        return HJGPU_OK;
    };
    if (c->debug_forensics >= 2) c->frozen.clear();
    std::vector<u64> pending;
    std::vector<std::vector<u64>> pending_pieces;
    std::vector<const void *> pending_base;
    int pending_slice = -1;
    for (int i = 0; i < slices; ++i) {
        const int slot = i & 1;
        for (int l = 0; l < L; ++l) {
            size_t b, e;
            range_of(shards[l].outer, 16, (size_t)i, (size_t)slices, &b, &e);
            in[l] = {shards[l].outer ? shards[l].d_outer_keys + b : nullptr, shards[l].outer ? shards[l].d_outer_vals + b : nullptr, e - b};
        }
        // partition(i) is enqueued first, then join(i-1) - which the device starts as soon as exchange(i-1) has arrived -
        // and only then does the host wait for partition(i)'s counts: the device works on join(i-1) and partition(i)
        // while the host and the ranks settle the sizes of exchange(i) (with the join enqueued after that wait, every
        // slice cost a world of one ~0.15-0.2 ms of idle device: 4 slices 13.0 -> 12.2-12.6 ms, 8 slices 16.7 -> 14.9-15.0 ms)
        // (host path with an event per uploaded slice: the one slice of a grouped join is the whole shard - its last upload)
        CHKM(step.begin_exchange(in, 1 + slot, slot, from_host ? &Rank::ev_up_s : nullptr,
                                 from_host && slice_events ? (grouped ? asked_slices - 1 : i) : -1));
        if (pending_slice >= 0) CHKM(join_slice(pending_slice, pending_slice & 1, pending, pending_pieces, pending_base));
        CHKM(step.finish_exchange(in, 1 + slot, slot));
        if (c->debug_forensics >= 2)
            for (int l = 0; l < L; ++l) {
                CHKM(freeze_check(l, 0, i));                            // partition(i): the host has just waited for its counts
                if (pending_slice >= 0) CHKM(freeze_check(l, 1, pending_slice));           // join(i - 1): waited for here; join(i) is not enqueued yet
            }
        if (build_stats_pending && i == 0) {
            // the host has just waited for this slice's partition counts and the counts gather; the build was enqueued
            // before both and is (nearly always) done: reading its events costs the pipeline nothing
            hjgpu_stats js;
            if (hjgpu_get_stats(c->ranks[0].join, &js) == HJGPU_OK) { add_stats(&stats->join, js); stats->joins += 1; stats->tuples_joined += inner_recv[0]; }
            build_stats_pending = false;
        }
        pending = step.recv_total;
        pending_pieces = step.pieces;
        pending_base = step.base;
        pending_slice = i;
    }
    CHKM(join_slice(pending_slice, pending_slice & 1, pending, pending_pieces, pending_base));
    if (c->debug_forensics >= 2) for (int l = 0; l < L; ++l) CHKM(freeze_check(l, 1, pending_slice));
    const int status = reduce_results(c, result);
    if (status != HJGPU_OK && status != HJGPU_EOVERFLOW) return status;
    if (c->debug_forensics) {
        // every stage's checksum of this step (the streams are idle: reduce_results waited for them)
        c->forensics.clear();
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            uint64_t end_part = 0, end_join = 0;
            (void)hjgpu_audit_read(r.part, &end_part, 0, 0, nullptr, nullptr);
            (void)hjgpu_audit_read(r.join, &end_join, 0, 0, nullptr, nullptr);
            const uint32_t np = (uint32_t)(end_part - seq_part[(size_t)l]), nj = (uint32_t)(end_join - seq_join[(size_t)l]);
            const size_t at = c->forensics.size();
            c->forensics.resize(at + 3 + ((size_t)np + nj) * 32, 0);
            c->forensics[at] = (u64)r.global; c->forensics[at + 1] = np; c->forensics[at + 2] = nj;
            JOINM(c, r.part, hjgpu_audit_read(r.part, nullptr, seq_part[(size_t)l], np, reinterpret_cast<uint64_t *>(c->forensics.data() + at + 3), r.prep));
            JOINM(c, r.join, hjgpu_audit_read(r.join, nullptr, seq_join[(size_t)l], nj, reinterpret_cast<uint64_t *>(c->forensics.data() + at + 3 + (size_t)np * 32), r.main));
        }
    }
    bool rows_fit = true;
    if (rows) for (int l = 0; l < L; ++l) {
        rows[l].rows = hp_local(c->ranks[l], G)[0];
        if (rows[l].rows > rows[l].out.capacity) rows_fit = false;
    }
    step.note_exchange();
    if (stats) {
        Rank &r = c->ranks[0];
        HIPM(c, hipSetDevice(r.device));
        hjgpu_stats js;
        if (measured && hjgpu_get_stats(r.join, &js) == HJGPU_OK) { add_stats(&stats->join, js); stats->joins += 1; stats->tuples_joined += measured; }
        for (int i = 0; i < slices; ++i) stats->ms_exchange_wait += elapsed(r.ev_w[2 * (size_t)i], r.ev_w[2 * (size_t)i + 1]);
        stats->ms_exchange = step.exchange_ms;
        stats->ms_wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    // belt and braces: whatever the flags say, more rows than a rank's columns hold is an overflow
    if (status == HJGPU_OK && !rows_fit)
        return cfail(c, HJGPU_EOVERFLOW, "materialised output exceeded the capacity of a local rank's result columns (rows needed per rank: hjgpu_shard_rows.rows)");
    return status;
}

int new_comm(int nranks, hjgpu_comm **out, hjgpu_comm **c)
{
    g_create_error[0] = 0;
    if (!out) return HJGPU_EINVAL;
    *out = nullptr;
    if (nranks < 1 || nranks > 1024) { snprintf(g_create_error, sizeof(g_create_error), "a communicator has 1 to 1024 ranks"); return HJGPU_EINVAL; }
    *c = new hjgpu_comm();
    (*c)->err[0] = 0;
    (*c)->why_broken[0] = 0;
    (*c)->nranks = nranks;
    const char *t = getenv("HJGPU_COMM_TIMEOUT_MS");       // read once per communicator, like the contexts' HJGPU_<OPTION>
    if (t && *t) { const long v = strtol(t, nullptr, 10); if (v > 0 && v < (1L << 30)) (*c)->timeout_ms = (int)v; }
    return HJGPU_OK;
}

// a communicator that cannot be made: its text survives it (hjgpu_comm_last_error(NULL) on this thread)
int fail_create(hjgpu_comm *c, int rc, const char *what = nullptr)
{
    if (what) snprintf(g_create_error, sizeof(g_create_error), "%s", what);
    else if (c && c->err[0]) snprintf(g_create_error, sizeof(g_create_error), "%s", c->err);
    else snprintf(g_create_error, sizeof(g_create_error), "%s", hjgpu_status_string(rc));
    if (c) hjgpu_comm_destroy(c);
    return rc;
}

// ---- preflight ----------------------------------------------------------------------------------------------
__global__ void pattern_kernel(u64 *__restrict__ p, u64 n, u64 seed)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        __builtin_nontemporal_store((u64)((seed + i) * 0x9E3779B97F4A7C15ull + (seed << 17)), &p[i]);
}
inline u64 pattern_at(u64 seed, u64 i) { return (seed + i) * 0x9E3779B97F4A7C15ull + (seed << 17); }

}  // namespace

// =====================================================================================================
extern "C" {

int hjgpu_comm_destroy(hjgpu_comm *c)
{
    if (!c) return HJGPU_OK;
    // RCCL first when the communicator gave up: its kernels may still sit on the streams, waiting for a peer
    if (c->broken) c->transport.reset();
    bool drained = true;
    if (!c->broken) for (Rank &r : c->ranks) { (void)hipSetDevice(r.device); (void)hipDeviceSynchronize(); }
    else {
        // a communicator that gave up: whatever is still on its streams may never finish (a kernel waiting for a peer
        // that is gone).  The streams are polled for at most the communicator's deadline (2 s without one); if they
        // have not drained by then the ranks' device resources are LEAKED rather than freed - hipFree, hipStreamDestroy
        // and hipDeviceSynchronize all wait for the device - so that "never a hang" holds for the clean-up too
        const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(c->timeout_ms > 0 ? c->timeout_ms : 2000);
        for (Rank &r : c->ranks) {
            (void)hipSetDevice(r.device);
            for (hipStream_t st : {r.up, r.prep, r.comm, r.main}) {
                if (!st) continue;
                hipError_t e;
                while ((e = hipStreamQuery(st)) == hipErrorNotReady && std::chrono::steady_clock::now() < until)
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                if (e == hipErrorNotReady) drained = false;
            }
            (void)hipGetLastError();
        }
    }
    c->transport.reset();                    // communicators before the streams they used
    if (drained) for (Rank &r : c->ranks) destroy_rank(r);
    delete c;
    return HJGPU_OK;
}

int hjgpu_comm_create_local(int nranks, const int *devices, int transport, hjgpu_comm **out)
{
    hjgpu_comm *c = nullptr;
    CHKM(new_comm(nranks, out, &c));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail_create(c, HJGPU_ENODEVICE, "no GPU device visible");
    c->first = 0;
    c->ranks.resize((size_t)nranks);
    int rc = HJGPU_OK;
    std::vector<int> devs;
    for (int i = 0; i < nranks && rc == HJGPU_OK; ++i) {
        const int d = devices ? devices[i] : (transport == HJGPU_TRANSPORT_LOOPBACK ? i % ndev : i);
        if (d < 0 || d >= ndev) { rc = cfail(c, HJGPU_EINVAL, "a rank names a device that is not visible"); break; }
        for (int q : devs) if (q == d && transport == HJGPU_TRANSPORT_RCCL) rc = cfail(c, HJGPU_EINVAL, "RCCL: one rank per device");
        devs.push_back(d);
        if (rc == HJGPU_OK) rc = init_rank(c, c->ranks[(size_t)i], d, i);
    }
    if (rc == HJGPU_OK) {
        if (transport == HJGPU_TRANSPORT_LOOPBACK) c->transport.reset(new LoopbackTransport(c));
        else if (transport == HJGPU_TRANSPORT_RCCL) {
            Rccl *R = rccl();
            if (!R) return fail_create(c, HJGPU_ERCCL, "RCCL transport: librccl.so.1 cannot be loaded (or lacks a symbol)");
            RcclTransport *t = new RcclTransport(c, R);
            c->transport.reset(t);
            t->comms.assign((size_t)nranks, nullptr);
            const ncclResult_t r = R->CommInitAll(t->comms.data(), nranks, devs.data());
            if (r != ncclSuccess) rc = cfail(c, HJGPU_ERCCL, "ncclCommInitAll", R->GetErrorString(r));
        } else rc = cfail(c, HJGPU_EINVAL, "unknown transport");
    }
    if (rc == HJGPU_OK) rc = apply_reserve(c);
    // ranks that share a device (loopback tests) share its free memory: their workspaces take the first allocation
    // instead of holding up to 12 candidates of the probe side's twin side by side (option "placement")
    for (size_t i = 0; i < devs.size() && rc == HJGPU_OK; ++i) {
        int sharers = 0;
        for (int d : devs) sharers += d == devs[i];
        if (sharers > 1 && (hjgpu_set_option(c->ranks[i].join, "placement", "1") != HJGPU_OK ||
                            hjgpu_set_option(c->ranks[i].part, "placement", "1") != HJGPU_OK)) rc = cfail(c, HJGPU_EINVAL, "placement");
    }
    if (rc != HJGPU_OK) return fail_create(c, rc);
    *out = c;
    return HJGPU_OK;
}

int hjgpu_comm_get_id(hjgpu_comm_id *id)
{
    static_assert(sizeof(ncclUniqueId) <= sizeof(hjgpu_comm_id), "hjgpu_comm_id too small for ncclUniqueId");
    if (!id) return HJGPU_EINVAL;
    memset(id, 0, sizeof(*id));
    Rccl *R = rccl();
    if (!R) { snprintf(g_create_error, sizeof(g_create_error), "librccl.so.1 cannot be loaded"); return HJGPU_ERCCL; }
    ncclUniqueId u;
    const ncclResult_t r = R->GetUniqueId(&u);
    if (r != ncclSuccess) { snprintf(g_create_error, sizeof(g_create_error), "ncclGetUniqueId: %s", R->GetErrorString(r)); return HJGPU_ERCCL; }
    memcpy(id->bytes, &u, sizeof(u));
    return HJGPU_OK;
}

int hjgpu_comm_create_rank(int device, int nranks, int rank, const hjgpu_comm_id *id, hjgpu_comm **out)
{
    hjgpu_comm *c = nullptr;
    CHKM(new_comm(nranks, out, &c));
    if (!id || rank < 0 || rank >= nranks) return fail_create(c, HJGPU_EINVAL, "hjgpu_comm_create_rank: null id or rank outside the world");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return fail_create(c, HJGPU_ENODEVICE, "no current device");
    c->first = rank;
    c->ranks.resize(1);
    int rc = init_rank(c, c->ranks[0], device, rank);
    if (rc == HJGPU_OK) {
        Rccl *R = rccl();
        if (!R) return fail_create(c, HJGPU_ERCCL, "RCCL transport: librccl.so.1 cannot be loaded (or lacks a symbol)");
        RcclTransport *t = new RcclTransport(c, R);
        c->transport.reset(t);
        t->comms.assign(1, nullptr);
        ncclUniqueId u;
        memcpy(&u, id->bytes, sizeof(u));
        if (hipSetDevice(device) != hipSuccess) rc = cfail(c, HJGPU_EHIP, "hipSetDevice");
        else {
            const ncclResult_t r = R->CommInitRank(&t->comms[0], nranks, u, rank);
            if (r != ncclSuccess) rc = cfail(c, HJGPU_ERCCL, "ncclCommInitRank", R->GetErrorString(r));
        }
    }
    if (rc == HJGPU_OK) rc = apply_reserve(c);
    if (rc != HJGPU_OK) return fail_create(c, rc);
    *out = c;
    return HJGPU_OK;
}

const char *hjgpu_comm_last_error(const hjgpu_comm *c) { return c ? c->err : g_create_error; }

int hjgpu_comm_size(const hjgpu_comm *c, int *nranks, int *nlocal, int *first_rank)
{
    if (!c) return HJGPU_EINVAL;
    if (nranks) *nranks = c->nranks;
    if (nlocal) *nlocal = (int)c->ranks.size();
    if (first_rank) *first_rank = c->first;
    return HJGPU_OK;
}

int hjgpu_comm_get_info(hjgpu_comm *c, hjgpu_comm_info *info)
{
    if (!c || !info) return HJGPU_EINVAL;
    memset(info, 0, sizeof(*info));
    info->nranks = c->nranks; info->nlocal = (int)c->ranks.size(); info->first_rank = c->first;
    info->rccl_nranks = info->rccl_rank = info->rccl_device = -1;
    info->timeout_ms = c->timeout_ms;
    info->aborted = c->broken ? 1 : 0;
    snprintf(info->transport, sizeof(info->transport), "%s", c->transport ? c->transport->name() : "none");
    if (c->transport && !c->broken) c->transport->info(info);
    return HJGPU_OK;
}

hjgpu_ctx *hjgpu_comm_ctx(hjgpu_comm *c, int local_rank)
{
    if (!c || local_rank < 0 || local_rank >= (int)c->ranks.size()) return nullptr;
    return c->ranks[(size_t)local_rank].join;
}

int hjgpu_comm_set_option(hjgpu_comm *c, const char *name, const char *value)
{
    if (!c || !name || !value) return HJGPU_EINVAL;
    char *end = nullptr;
    const long long x = strtoll(value, &end, 10);
    if (end == value || *end) return cfail(c, HJGPU_EINVAL, "hjgpu_comm_set_option: malformed value");
    if (strcmp(name, "ring_broadcast") == 0) { c->ring_broadcast = x != 0; return HJGPU_OK; }
    if (strcmp(name, "reserve_cus") == 0) {
        if (x < 0 || x > 128) return cfail(c, HJGPU_EINVAL, "hjgpu_comm_set_option: reserve_cus outside 0..128");
        c->reserve_cus = (int)x;
        return apply_reserve(c);
    }
    if (strcmp(name, "max_message_bytes") == 0) {
        if (x < 16) return cfail(c, HJGPU_EINVAL, "hjgpu_comm_set_option: max_message_bytes below 16");
        c->max_message_bytes = (size_t)x;
        return HJGPU_OK;
    }
    if (strcmp(name, "exchange_in_place") == 0) { c->exchange_in_place = x != 0; return HJGPU_OK; }
    if (strcmp(name, "cpra_k") == 0) { if (x < 0 || x > 192) return cfail(c, HJGPU_EINVAL, "cpra_k: 0 ... 192"); c->cpra_k = (int)x; return HJGPU_OK; }
    if (strcmp(name, "cpra_two_level") == 0) { c->cpra_two_level = x != 0; return HJGPU_OK; }
    if (strcmp(name, "cpra_grouped") == 0) { if (x < 0 || x > 2) return cfail(c, HJGPU_EINVAL, "cpra_grouped: 0, 1 or 2"); c->cpra_grouped = (int)x; return HJGPU_OK; }
    if (strcmp(name, "cpra_fused_counts") == 0) { c->cpra_fused_counts = x != 0; return HJGPU_OK; }
    if (strcmp(name, "host_rows_batched") == 0) { c->host_rows_batched = x != 0; return HJGPU_OK; }
    if (strcmp(name, "debug_serialize") == 0) { c->debug_serialize = (int)x; return HJGPU_OK; }
    if (strcmp(name, "debug_forensics") == 0) {
        for (Rank &r : c->ranks)
            for (hjgpu_ctx *ctx : {r.join, r.part})
                if (ctx && hjgpu_set_option(ctx, "audit", x ? "1" : "0") != HJGPU_OK) return cfail(c, HJGPU_EINVAL, "debug_forensics: option audit");
        c->debug_forensics = (int)x;
        return HJGPU_OK;
    }
    if (strcmp(name, "self_via_rccl") == 0) { c->self_via_rccl = x != 0; return HJGPU_OK; }
    if (strcmp(name, "timeout_ms") == 0) {
        if (x < 0 || x > (1 << 30)) return cfail(c, HJGPU_EINVAL, "hjgpu_comm_set_option: timeout_ms outside 0..2^30");
        c->timeout_ms = (int)x;
        return HJGPU_OK;
    }
    if (strcmp(name, "stall_rank") == 0) {                  // fault injection, loopback transport only
        if (!c->transport || strcmp(c->transport->name(), "loopback") != 0) return cfail(c, HJGPU_EINVAL, "hjgpu_comm_set_option: stall_rank is a loopback test switch");
        c->stall_rank = (int)x;
        return HJGPU_OK;
    }
    if (strcmp(name, "stall_ms") == 0) {
        if (x < 0 || x > 10000) return cfail(c, HJGPU_EINVAL, "hjgpu_comm_set_option: stall_ms outside 0..10000");
        c->stall_ms = (int)x;
        return HJGPU_OK;
    }
    return cfail(c, HJGPU_EINVAL, "hjgpu_comm_set_option: unknown option");
}

// hjgpu_audit_recheck on both contexts of every local rank (option "debug_forensics"): words = for every local rank and context
// {global rank, context (0 partitioning, 1 join), n} followed by n x 9 words
int hjgpu_comm_recheck(hjgpu_comm *c, uint64_t *words, size_t capacity, size_t *count)
{
    if (!c || !count) return HJGPU_EINVAL;
    std::vector<u64> all;
    for (auto &r : c->ranks) {
        hjgpu_ctx *both[2] = {r.part, r.join};
        for (int which = 0; which < 2; ++which) {
            size_t n = 0;
            if (!both[which]) continue;
            JOINM(c, both[which], hjgpu_audit_recheck(both[which], nullptr, 0, &n));
            const size_t at = all.size();
            all.resize(at + 3 + 9 * n, 0);
            all[at] = (u64)r.global; all[at + 1] = (u64)which; all[at + 2] = (u64)n;
            if (n) JOINM(c, both[which], hjgpu_audit_recheck(both[which], reinterpret_cast<uint64_t *>(all.data() + at + 3), n, &n));
        }
    }
    *count = all.size();
    if (!words || capacity < all.size()) return HJGPU_OK;
    if (!all.empty()) memcpy(words, all.data(), all.size() * sizeof(u64));
    return HJGPU_OK;
}

// option "debug_forensics" = 2: the stages of the last hjgpu_cpra_multi step that were found wrong while their buffers were still intact, each
// looked at again on the spot: per event {global rank, context (0 partitioning, 1 join), slice, n, the call's record (32 words)} + n x 9 words
// (hjgpu_audit_recheck)
int hjgpu_comm_get_frozen(hjgpu_comm *c, uint64_t *words, size_t capacity, size_t *count)
{
    if (!c || !count) return HJGPU_EINVAL;
    *count = c->frozen.size();
    if (!words || capacity < c->frozen.size()) return HJGPU_OK;
    if (!c->frozen.empty()) memcpy(words, c->frozen.data(), c->frozen.size() * sizeof(u64));
    return HJGPU_OK;
}

int hjgpu_comm_get_forensics(hjgpu_comm *c, uint64_t *words, size_t capacity, size_t *count)
{
    if (!c || !count) return HJGPU_EINVAL;
    *count = c->forensics.size();
    if (!words || capacity < c->forensics.size()) return HJGPU_OK;       // the caller asks again with room for *count words
    if (!c->forensics.empty()) memcpy(words, c->forensics.data(), c->forensics.size() * sizeof(u64));
    return HJGPU_OK;
}

int hjgpu_comm_barrier(hjgpu_comm *c)
{
    if (!c) return HJGPU_EINVAL;
    CHKM(refuse_broken(c));
    CHKM(sync_all(c));
    std::vector<u64 *> one;
    for (Rank &r : c->ranks) one.push_back(static_cast<u64 *>(r.d_cnt.p));
    const std::vector<hipStream_t> comms = streams_of(c, &Rank::comm);
    CHKM(c->transport->all_reduce_u64(one.data(), 1, comms.data()));
    return sync_all(c);
}

// Checksum-verified collectives and point-to-point rates BEFORE a join is trusted to the links (SURVEY section 5:
// "measure link bandwidth first"): 1 MB through every collective the joins use, verified word for word on the host;
// then `link_bytes` from every rank to the peer k places on, k = 1 .. G-1 (all ranks at once: every pair's own
// link), and finally to all peers at once (the all-to-all-v shape of the CPRA exchange).
int hjgpu_comm_preflight(hjgpu_comm *c, size_t link_bytes, hjgpu_preflight *rep)
{
    if (!c || !rep) return HJGPU_EINVAL;
    CHKM(refuse_broken(c));
    memset(rep, 0, sizeof(*rep));
    const int L = (int)c->ranks.size(), G = c->nranks;
    const size_t Gs = (size_t)G;
    if (G > (int)(sizeof(rep->link_GBs) / sizeof(rep->link_GBs[0]))) return cfail(c, HJGPU_EINVAL, "preflight reports at most 64 ranks");
    rep->nranks = (uint32_t)G; rep->rank = (uint32_t)c->ranks[0].global;
    const size_t W = 131072;                                   // 1 MB of u64 per rank and collective
    link_bytes = (link_bytes + 15) & ~size_t(15);
    rep->link_bytes = link_bytes;
    // scratch per rank: [W] send | [G * W] all-gather | [G * W] all-to-all send | [G * W] all-to-all receive | [W] reduce
    //                   | link send [link_bytes] | link receive [G * link_bytes]
    const size_t words = W + 3 * Gs * W + W;
    std::vector<u64 *> base(L);
    std::vector<std::vector<u64>> host(L);
    const std::vector<hipStream_t> comms = streams_of(c, &Rank::comm);
    for (int l = 0; l < L; ++l) {
        Rank &r = c->ranks[l];
        CHKM(ensure(c, r, r.pre, words * sizeof(u64) + (Gs + 1) * link_bytes + 256));
        base[l] = static_cast<u64 *>(r.pre.p);
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hj_zero_async(base[l], words * sizeof(u64), r.comm));
        hipLaunchKernelGGL(pattern_kernel, dim3(64), dim3(256), 0, r.comm, base[l], (u64)W, (u64)(1000 + r.global));                       // all-gather
        hipLaunchKernelGGL(pattern_kernel, dim3(256), dim3(256), 0, r.comm, base[l] + W + Gs * W, (u64)(Gs * W), (u64)(5000 + r.global));   // all-to-all
        hipLaunchKernelGGL(pattern_kernel, dim3(64), dim3(256), 0, r.comm, base[l] + W + 3 * Gs * W, (u64)W, (u64)(9000 + r.global));       // all-reduce
        HIPM(c, hipGetLastError());
        host[l].resize(words);
    }
    auto timed = [&](float *ms, auto enqueue) -> int {
        Rank &r0 = c->ranks[0];
        HIPM(c, hipSetDevice(r0.device));
        HIPM(c, hipEventRecord(r0.ev_x0, r0.comm));
        CHKM(enqueue());
        HIPM(c, hipSetDevice(r0.device));
        HIPM(c, hipEventRecord(r0.ev_x1, r0.comm));
        CHKM(sync_all(c));
        HIPM(c, hipSetDevice(r0.device));
        *ms = elapsed(r0.ev_x0, r0.ev_x1);
        return HJGPU_OK;
    };
    // ---- all-gather
    {
        std::vector<const void *> s; std::vector<void *> d;
        for (int l = 0; l < L; ++l) { s.push_back(base[l]); d.push_back(base[l] + W); }
        CHKM(timed(&rep->ms_all_gather, [&] { return c->transport->all_gather(s.data(), d.data(), W * sizeof(u64), comms.data()); }));
    }
    // ---- all-to-all-v: rank s sends W - 37 * s - 11 * d words to rank d (different for every pair)
    std::vector<std::vector<u64>> soff(L, std::vector<u64>(Gs)), scnt = soff, roff = soff, rcnt = soff;
    auto a2a_count = [&](int s, int d) -> u64 { return (u64)W - 37ull * (u64)s - 11ull * (u64)d; };
    {
        std::vector<const void *> s; std::vector<void *> d;
        std::vector<const u64 *> so, sc, ro, rc;
        for (int l = 0; l < L; ++l) {
            const int me = c->ranks[l].global;
            for (int p = 0; p < G; ++p) {
                soff[l][p] = (u64)p * W; scnt[l][p] = a2a_count(me, p);
                roff[l][p] = (u64)p * W; rcnt[l][p] = a2a_count(p, me);
            }
            s.push_back(base[l] + W + Gs * W); d.push_back(base[l] + W + 2 * Gs * W);
            so.push_back(soff[l].data()); sc.push_back(scnt[l].data()); ro.push_back(roff[l].data()); rc.push_back(rcnt[l].data());
        }
        CHKM(timed(&rep->ms_all_to_all, [&] { return c->transport->all_to_all_v(s.data(), so.data(), sc.data(), d.data(), ro.data(), rc.data(), sizeof(u64), comms.data()); }));
    }
    // ---- all-reduce
    {
        std::vector<u64 *> b;
        for (int l = 0; l < L; ++l) b.push_back(base[l] + W + 3 * Gs * W);
        CHKM(timed(&rep->ms_all_reduce, [&] { return c->transport->all_reduce_u64(b.data(), W, comms.data()); }));
    }
    // ---- verify on the host
    for (int l = 0; l < L; ++l) {
        Rank &r = c->ranks[l];
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hipMemcpyAsync(host[l].data(), base[l], words * sizeof(u64), hipMemcpyDeviceToHost, r.comm));
    }
    CHKM(sync_all(c));
    bool ok_g = true, ok_a = true, ok_r = true;
    for (int l = 0; l < L; ++l) {
        const int me = c->ranks[l].global;
        const u64 *h = host[l].data();
        for (int p = 0; p < G && ok_g; ++p)
            for (size_t i = 0; i < W; ++i) if (h[W + (size_t)p * W + i] != pattern_at(1000 + (u64)p, i)) { ok_g = false; break; }
        for (int p = 0; p < G && ok_a; ++p) {
            const u64 n = a2a_count(p, me);
            // rank p filled its send buffer [G * W] with pattern(5000 + p); what it sends to me starts at word me * W
            for (u64 i = 0; i < n; ++i) if (h[W + 2 * Gs * W + (size_t)p * W + i] != pattern_at(5000 + (u64)p, (u64)me * W + i)) { ok_a = false; break; }
            if (n < W && h[W + 2 * Gs * W + (size_t)p * W + n] != 0) ok_a = false;          // nothing beyond the message
        }
        for (size_t i = 0; i < W && ok_r; ++i) {
            u64 want = 0;
            for (int p = 0; p < G; ++p) want += pattern_at(9000 + (u64)p, i);
            if (h[W + 3 * Gs * W + i] != want) ok_r = false;
        }
    }
    rep->ok_all_gather = ok_g; rep->ok_all_to_all = ok_a; rep->ok_all_reduce = ok_r;
    // ---- point-to-point rates: shift by k (every rank sends to the peer k places on), then all peers at once
    if (link_bytes && G > 1) {
        std::vector<const void *> s; std::vector<void *> d;
        std::vector<const u64 *> so, sc, ro, rc;
        for (int l = 0; l < L; ++l) {
            char *lb = reinterpret_cast<char *>(base[l] + words);
            s.push_back(lb); d.push_back(lb + link_bytes);
            so.push_back(soff[l].data()); sc.push_back(scnt[l].data()); ro.push_back(roff[l].data()); rc.push_back(rcnt[l].data());
        }
        for (int k = 1; k <= G; ++k) {                       // k == G: all peers at once
            for (int l = 0; l < L; ++l) {
                const int me = c->ranks[l].global;
                for (int p = 0; p < G; ++p) {
                    const bool to = k < G ? p == (me + k) % G : p != me, from = k < G ? p == (me - k + G) % G : p != me;
                    soff[l][p] = 0; scnt[l][p] = to ? link_bytes : 0;
                    roff[l][p] = (u64)p * link_bytes; rcnt[l][p] = from ? link_bytes : 0;
                }
            }
            float ms = 0;
            for (int rep_i = 0; rep_i < 2; ++rep_i)           // the first message over a link sets the connection up
                CHKM(timed(&ms, [&] { return c->transport->all_to_all_v(s.data(), so.data(), sc.data(), d.data(), ro.data(), rc.data(), 1, comms.data()); }));
            const float gbs = ms > 0 ? (float)((double)link_bytes / (ms * 1e-3) / 1e9) : 0.f;
            if (k < G) rep->link_GBs[(c->ranks[0].global + k) % G] = gbs;
            else rep->all_to_all_GBs = gbs * (float)(G - 1);       // sent to all peers together, per rank
        }
    }
    if (!(ok_g && ok_a && ok_r)) {
        char text[200];
        snprintf(text, sizeof(text), "preflight: corrupted collective (all-gather %s, all-to-all-v %s, all-reduce %s) over %s",
                 ok_g ? "ok" : "BAD", ok_a ? "ok" : "BAD", ok_r ? "ok" : "BAD", c->transport->name());
        return cfail(c, HJGPU_ERCCL, text);
    }
    return HJGPU_OK;
}

int hjgpu_phj_multi(hjgpu_comm *c, const hjgpu_shard *shards, int root, const hjgpu_phj_params *params,
                    hjgpu_result *result, hjgpu_multi_stats *stats)
{
    return replicated_join(c, 1, shards, nullptr, root, params, nullptr, result, stats);
}

int hjgpu_npj_multi(hjgpu_comm *c, const hjgpu_shard *shards, int root, const hjgpu_npj_params *params,
                    hjgpu_result *result, hjgpu_multi_stats *stats)
{
    return replicated_join(c, 0, shards, nullptr, root, nullptr, params, result, stats);
}

int hjgpu_cpra_multi(hjgpu_comm *c, const hjgpu_shard *shards, const hjgpu_phj_params *params, int slices,
                     hjgpu_result *result, hjgpu_multi_stats *stats)
{
    return cpra_join(c, shards, nullptr, params, slices, result, stats);
}

int hjgpu_phj_multi_rows(hjgpu_comm *c, const hjgpu_shard *shards, hjgpu_shard_rows *rows, int root,
                         const hjgpu_phj_params *params, hjgpu_result *result, hjgpu_multi_stats *stats)
{
    if (!rows) return cfail(c, HJGPU_EINVAL, "hjgpu_phj_multi_rows: rows is required");
    return replicated_join(c, 1, shards, rows, root, params, nullptr, result, stats);
}

int hjgpu_npj_multi_rows(hjgpu_comm *c, const hjgpu_shard *shards, hjgpu_shard_rows *rows, int root,
                         const hjgpu_npj_params *params, hjgpu_result *result, hjgpu_multi_stats *stats)
{
    if (!rows) return cfail(c, HJGPU_EINVAL, "hjgpu_npj_multi_rows: rows is required");
    return replicated_join(c, 0, shards, rows, root, nullptr, params, result, stats);
}

int hjgpu_cpra_multi_rows(hjgpu_comm *c, const hjgpu_shard *shards, hjgpu_shard_rows *rows, const hjgpu_phj_params *params,
                          int slices, hjgpu_result *result, hjgpu_multi_stats *stats)
{
    if (!rows) return cfail(c, HJGPU_EINVAL, "hjgpu_cpra_multi_rows: rows is required");
    return cpra_join(c, shards, rows, params, slices, result, stats);
}

// The host columns are cut into the ranks' shares; every rank's share travels on the rank's own upload stream into
// buffers that the communicator keeps between calls (no hipMalloc / hipFree per call), probe side first, and the
// joins are enqueued right behind: a rank partitions its probe shard while its (and the root's build) columns are
// still arriving - the single-GPU host path's pipeline (hjgpu_join_host), per rank.
// probe slices of the CPRA host call: the shard is uploaded, partitioned, exchanged and joined in this many pieces (a slice costs
// ~0.6 ms of launches; the upload of 1 / 8 of a shard takes ~10 ms at the PCIe rate)
constexpr int HOST_CPRA_SLICES = 8;

static int join_host_multi_impl(hjgpu_comm *c, int algorithm,
                                const uint32_t *ik, const uint32_t *iv, size_t inner,
                                const uint32_t *ok, const uint32_t *ov, size_t outer,
                                const hjgpu_phj_params *pp, const hjgpu_npj_params *np,
                                const hjgpu_host_rows *host_rows, hjgpu_result *result, hjgpu_multi_stats *stats)
{
    if (!c || algorithm < 0 || algorithm > 2) return HJGPU_EINVAL;
    CHKM(refuse_broken(c));
    if ((int)c->ranks.size() != c->nranks) return cfail(c, HJGPU_EINVAL, "hjgpu_join_host_multi needs a local communicator");
    if ((inner && (!ik || !iv)) || (outer && (!ok || !ov))) return cfail(c, HJGPU_EINVAL, "null column");
    if (host_rows && (!result || (host_rows->capacity && (!host_rows->keys || !host_rows->outer_vals || !host_rows->inner_vals))))
        return cfail(c, HJGPU_EINVAL, "hjgpu_join_host_rows_multi: result and the three result columns are required");
    const int G = c->nranks;
    const auto t0 = std::chrono::steady_clock::now();
    if (host_rows && algorithm != 2 && c->host_rows_batched && inner && outer) {
        // Replicated build side with rows: every rank runs the ONE-GPU host pipeline on its probe shard (npj.cpp:1013-1039
        // reads the columns, 997-1000 sizes the output: here per GPU) - the build side read from the host by every GPU
        // (SURVEY 8e), the shard in batches behind the DMA, a batch's dense rows going home while the next is joined and the
        // one after it uploaded - and all ranks append to the caller's columns through one atomic cursor: the concatenation
        // is the result, in the order the batches finish.  A rank whose rows do not fit (a skewed batch, or the capacity)
        // counts them: the call then reports the needed rows, or starts over on the whole-shard path below.
        uint64_t cursor = 0;
        std::vector<hjgpu_result> res((size_t)G);
        std::vector<hjgpu_stats> st((size_t)G);
        std::vector<int> rcs((size_t)G, HJGPU_OK);
        (void)each_rank(G, [&](int g) -> int {
            Rank &r = c->ranks[(size_t)g];
            size_t sb, se;
            range_of(outer, 16, (size_t)g, (size_t)G, &sb, &se);
            memset(&res[(size_t)g], 0, sizeof(hjgpu_result));
            rcs[(size_t)g] = hjgpu_join_host_rows_shared(r.join, algorithm, ik, iv, inner, ok + sb, ov + sb, se - sb, pp, np, host_rows, &cursor,
                                                        &res[(size_t)g], &st[(size_t)g]);
            return HJGPU_OK;
        });
        bool overflow = false;
        hjgpu_result sum;
        memset(&sum, 0, sizeof(sum));
        for (int g = 0; g < G; ++g) {
            const int rg = rcs[(size_t)g];
            if (rg == HJGPU_EOVERFLOW) overflow = true;
            else if (rg != HJGPU_OK) return cfail(c, rg, "hjgpu_join_host_rows_shared", hjgpu_last_error(c->ranks[(size_t)g].join));
            sum.count += res[(size_t)g].count; sum.sum_keys += res[(size_t)g].sum_keys;
            sum.sum_outer_vals += res[(size_t)g].sum_outer_vals; sum.sum_inner_vals += res[(size_t)g].sum_inner_vals;
        }
        if (!overflow || sum.count > host_rows->capacity) {
            *result = sum;
            if (stats) {
                memset(stats, 0, sizeof(*stats));
                stats->join = st[0]; stats->ms_upload = st[0].ms_upload; stats->joins = st[0].batches ? st[0].batches : 1;
                stats->tuples_joined = inner + outer / (size_t)G;
                stats->ms_wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            if (overflow) return cfail(c, HJGPU_EOVERFLOW, "hjgpu_join_host_rows_multi: the result has more rows than rows->capacity (see result->count)");
            return HJGPU_OK;
        }
        // a batch outgrew its device columns although the whole fits the capacity (skew): whole shards, below
    }
    std::vector<hjgpu_shard> shards((size_t)G);
    for (int g = 0; g < G; ++g) {
        Rank &r = c->ranks[(size_t)g];
        hjgpu_shard &s = shards[(size_t)g];
        memset(&s, 0, sizeof(s));
        size_t sb, se, rb = 0, re = 0;
        range_of(outer, 16, (size_t)g, (size_t)G, &sb, &se);            // thread_beg / thread_end with T = ranks
        s.outer = se - sb;
        if (algorithm == 2) { range_of(inner, 16, (size_t)g, (size_t)G, &rb, &re); s.inner = re - rb; }
        else { s.inner = inner; if (g == 0) { rb = 0; re = inner; } }
        const uint32_t *h[4] = {ik + rb, iv + rb, ok + sb, ov + sb};
        const size_t n[4] = {re - rb, re - rb, s.outer, s.outer};
        for (int i = 0; i < 4; ++i) CHKM(ensure(c, r, r.shard[i], (n[i] + 4) * sizeof(uint32_t)));
        HIPM(c, hipSetDevice(r.device));
        // pinned columns (hjgpu_host_alloc) are DMA'd; the GPUs' uploads run side by side
        if (algorithm == 2) {
            // CPRA needs the build side first (its exchange and the prepared build come before any probe slice), then the
            // probe shard arrives in the slices the join takes it in: slice i is partitioned, exchanged and joined while the
            // later slices are still on the bus (the reference reads all four columns before its clock starts,
            // cpra2.cpp:2110-2136; here the upload is most of a call: 8.5 GB at the PCIe rate against ~15 ms of join)
            while (r.ev_up_slice.size() < (size_t)HOST_CPRA_SLICES) {
                hipEvent_t e = nullptr;
                HIPM(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                r.ev_up_slice.push_back(e);
            }
            for (int i : {0, 1}) if (n[i]) HIPM(c, hipMemcpyAsync(r.shard[i].p, h[i], n[i] * sizeof(uint32_t), hipMemcpyHostToDevice, r.up));
            HIPM(c, hipEventRecord(r.ev_up_r, r.up));
            for (int i = 0; i < HOST_CPRA_SLICES; ++i) {
                size_t b, e;
                range_of(s.outer, 16, (size_t)i, (size_t)HOST_CPRA_SLICES, &b, &e);       // the slices of cpra_join
                for (int col : {2, 3})
                    if (e > b) HIPM(c, hipMemcpyAsync(static_cast<uint32_t *>(r.shard[col].p) + b, h[col] + b, (e - b) * sizeof(uint32_t), hipMemcpyHostToDevice, r.up));
                HIPM(c, hipEventRecord(r.ev_up_slice[(size_t)i], r.up));
            }
            HIPM(c, hipEventRecord(r.ev_up_s, r.up));
        } else {
            // PHJ / NPJ: probe side first - it is partitioned while the build side is still arriving
            for (int i : {2, 3, 0, 1}) {
                if (n[i]) HIPM(c, hipMemcpyAsync(r.shard[i].p, h[i], n[i] * sizeof(uint32_t), hipMemcpyHostToDevice, r.up));
                if (i == 3) HIPM(c, hipEventRecord(r.ev_up_s, r.up));
            }
            HIPM(c, hipEventRecord(r.ev_up_r, r.up));
        }
        HIPM(c, hipEventRecord(r.ev_t1, r.up));
        s.d_outer_keys = static_cast<const uint32_t *>(r.shard[2].p); s.d_outer_vals = static_cast<const uint32_t *>(r.shard[3].p);
        if (re > rb) { s.d_inner_keys = static_cast<const uint32_t *>(r.shard[0].p); s.d_inner_vals = static_cast<const uint32_t *>(r.shard[1].p); }
    }
    std::vector<hjgpu_shard_rows> rows;
    int rc = HJGPU_OK;
    if (host_rows) {
        // result columns per rank: its share of the caller's capacity with a quarter of headroom; a rank that needs
        // more says how much (rows[l].rows) and the join is run once more with exactly that
        rows.resize((size_t)G);
        const size_t bs = host_rows->capacity >= (64u << 20) ? 65536 : 1024;
        for (int attempt = 0; attempt < 2; ++attempt) {
            for (int g = 0; g < G; ++g) {
                Rank &r = c->ranks[(size_t)g];
                const size_t want_rows = attempt ? (size_t)rows[(size_t)g].rows : host_rows->capacity / (size_t)G + host_rows->capacity / (size_t)(4 * G) + 1;
                size_t cap = 0;
                JOINM(c, r.join, hjgpu_output_capacity(r.join, algorithm, shards[(size_t)g].outer, want_rows, bs, &cap));
                for (int i = 0; i < 3; ++i) CHKM(ensure(c, r, r.rows_col[i], cap * sizeof(uint32_t)));
                hjgpu_output &o = rows[(size_t)g].out;
                o.d_keys = static_cast<uint32_t *>(r.rows_col[0].p); o.d_outer_vals = static_cast<uint32_t *>(r.rows_col[1].p);
                o.d_inner_vals = static_cast<uint32_t *>(r.rows_col[2].p);
                o.capacity = cap; o.block_size = bs;
            }
            if (algorithm == 2) rc = cpra_join(c, shards.data(), rows.data(), pp, HOST_CPRA_SLICES, result, stats, attempt == 0, attempt == 0);
            else rc = replicated_join(c, algorithm, shards.data(), rows.data(), 0, pp, np, result, stats, attempt == 0);
            if (rc != HJGPU_EOVERFLOW) break;
        }
    } else {
        if (algorithm == 2) rc = cpra_join(c, shards.data(), nullptr, pp, HOST_CPRA_SLICES, result, stats, true, true);
        else rc = replicated_join(c, algorithm, shards.data(), nullptr, 0, pp, np, result, stats, true);
    }
    if (rc != HJGPU_OK) { if (!c->broken) (void)sync_all(c); return rc; }
    if (stats) {
        // local rank 0: how long its columns took to arrive, and whether its first join kernel started before that
        Rank &r = c->ranks[0];
        HIPM(c, hipSetDevice(r.device));
        stats->ms_upload = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() - stats->ms_wall;
        if (stats->ms_upload < 0) stats->ms_upload = 0;
        // > 0: the join's first kernel was on the device this long BEFORE the last byte of the upload arrived
        stats->ms_overlap = elapsed(r.ev_t0, r.ev_t1);
    }
    if (host_rows) {
        if (result->count > host_rows->capacity)
            return cfail(c, HJGPU_EOVERFLOW, "hjgpu_join_host_rows_multi: the result has more rows than rows->capacity (see result->count)");
        // concatenation = result: rank g's rows follow the rows of the ranks before it
        const auto d0 = std::chrono::steady_clock::now();
        u64 at = 0;
        for (int g = 0; g < G; ++g) {
            Rank &r = c->ranks[(size_t)g];
            const u64 n = rows[(size_t)g].rows;
            uint32_t *hcol[3] = {host_rows->keys, host_rows->outer_vals, host_rows->inner_vals};
            HIPM(c, hipSetDevice(r.device));
            for (int i = 0; i < 3 && n; ++i)
                HIPM(c, hipMemcpyAsync(hcol[i] + at, r.rows_col[i].p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, r.up));
            at += n;
        }
        CHKM(sync_all(c));
        if (at != result->count) return cfail(c, HJGPU_EHIP, "internal: the ranks' rows do not add up to the global count");
        if (stats) stats->join.ms_download = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - d0).count();
    }
    return HJGPU_OK;
}

int hjgpu_join_host_multi(hjgpu_comm *c, int algorithm,
                          const uint32_t *ik, const uint32_t *iv, size_t inner,
                          const uint32_t *ok, const uint32_t *ov, size_t outer,
                          const hjgpu_phj_params *pp, const hjgpu_npj_params *np,
                          hjgpu_result *result, hjgpu_multi_stats *stats)
{
    return join_host_multi_impl(c, algorithm, ik, iv, inner, ok, ov, outer, pp, np, nullptr, result, stats);
}

int hjgpu_join_host_rows_multi(hjgpu_comm *c, int algorithm,
                               const uint32_t *ik, const uint32_t *iv, size_t inner,
                               const uint32_t *ok, const uint32_t *ov, size_t outer,
                               const hjgpu_phj_params *pp, const hjgpu_npj_params *np,
                               const hjgpu_host_rows *rows, hjgpu_result *result, hjgpu_multi_stats *stats)
{
    if (!c) return HJGPU_EINVAL;
    if (!rows) return cfail(c, HJGPU_EINVAL, "hjgpu_join_host_rows_multi: rows is required");
    return join_host_multi_impl(c, algorithm, ik, iv, inner, ok, ov, outer, pp, np, rows, result, stats);
}

}  // extern "C"
