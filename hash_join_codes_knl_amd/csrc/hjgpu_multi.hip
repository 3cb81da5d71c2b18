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
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <memory>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/hjgpu.h"

typedef unsigned long long u64;

namespace {

const uint32_t TOP_LEVEL_FACTOR = 0x2C1B3C6Du;        // odd multiplier of the exchange-level partitioning

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
    hipEvent_t ev_ready = nullptr;  // build side replicated (PHJ / NPJ)
    hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;      // timing of an exchange on `comm`
    hipEvent_t ev_w0 = nullptr, ev_w1 = nullptr;      // timing of a join's wait for its exchange on `main`
    hipEvent_t ev_part[2] = {nullptr, nullptr};       // send buffers of slot b partitioned
    hipEvent_t ev_xchg[2] = {nullptr, nullptr};       // receive buffers of slot b filled
    hipEvent_t ev_join[2] = {nullptr, nullptr};       // receive buffers of slot b joined (free again)
    hipEvent_t ev_rx = nullptr;                       // build side received (CPRA)
    hipEvent_t lb_in = nullptr, lb_out = nullptr;     // loopback transport
    Buf rbuf;                       // PHJ / NPJ: replicated build side (keys | payloads)
    Buf send_k[2], send_v[2], recv_k[2], recv_v[2];   // CPRA: probe-side slices, double-buffered
    Buf rsend_k, rsend_v, rrecv_k, rrecv_v;           // CPRA: build side
    Buf d_off;                      // [2][G + 1] u64: partition offsets of slot b
    Buf d_cnt;                      // [G] u64 send counts | [G * G] gathered matrix
    Buf d_res;                      // [8] u64: accumulated result | last batch
    Buf scratch;                    // loopback all-reduce staging
    u64 *h_pin = nullptr;           // pinned host scratch: [2][G + 1] offsets | [G * G] matrix | [8] result | [G] send counts
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
    char err[512];
};

namespace {

int cfail(hjgpu_comm *c, int status, const char *what, const char *detail = nullptr)
{
    if (c) snprintf(c->err, sizeof(c->err), detail ? "%s: %s" : "%s", what, detail);
    return status;
}

#define HIPM(c, call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return cfail((c), HJGPU_EHIP, #call, hipGetErrorString(e_));   \
    } while (0)
#define NCCLM(c, call)                                                                       \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) return cfail((c), HJGPU_ERCCL, #call, ncclGetErrorString(r_)); \
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
    size_t replicate_capacity(size_t bytes) const
    {
        const size_t per = ((bytes + c->nranks - 1) / c->nranks + 15) & ~size_t(15);
        return per * c->nranks;
    }
};

// ---- RCCL over xGMI ---------------------------------------------------------------------------------
struct RcclTransport : Transport {
    std::vector<ncclComm_t> comms;           // one per local rank
    explicit RcclTransport(hjgpu_comm *comm) : Transport(comm) {}
    ~RcclTransport() override
    {
        for (size_t l = 0; l < comms.size(); ++l)
            if (comms[l]) { (void)hipSetDevice(c->ranks[l].device); (void)ncclCommDestroy(comms[l]); }
    }
    const char *name() const override { return "rccl"; }

    int all_gather(const void *const *send, void *const *recv, size_t bytes, hipStream_t const *streams) override
    {
        NCCLM(c, ncclGroupStart());
        for (int l = 0; l < nlocal(); ++l) {
            HIPM(c, hipSetDevice(c->ranks[l].device));
            NCCLM(c, ncclAllGather(send[l], recv[l], bytes, ncclUint8, comms[l], streams[l]));
        }
        NCCLM(c, ncclGroupEnd());
        return HJGPU_OK;
    }

    int replicate(void *const *bufs, size_t bytes, int root, bool ring, hipStream_t const *streams) override
    {
        const int G = c->nranks;
        if (G == 1) return HJGPU_OK;
        if (ring || bytes < (size_t)G * 4096) {
            // one ring broadcast: bound by ONE xGMI link (512 MB of build side at ~60 GB/s per direction = 8.5 ms)
            NCCLM(c, ncclGroupStart());
            for (int l = 0; l < nlocal(); ++l) {
                HIPM(c, hipSetDevice(c->ranks[l].device));
                NCCLM(c, ncclBroadcast(bufs[l], bufs[l], bytes, ncclUint8, root, comms[l], streams[l]));
            }
            NCCLM(c, ncclGroupEnd());
            return HJGPU_OK;
        }
        // xGMI is point to point (7 links per GPU, every pair directly connected): the root sends a DIFFERENT
        // 1/G slice to every peer over its own link, then everybody exchanges slices - all links of all GPUs
        // carry data, 2 x ~1/(G-1) of the single-link time
        const size_t per = replicate_capacity(bytes) / G;
        NCCLM(c, ncclGroupStart());
        for (int l = 0; l < nlocal(); ++l) {
            const int g = c->ranks[l].global;
            HIPM(c, hipSetDevice(c->ranks[l].device));
            char *b = static_cast<char *>(bufs[l]);
            if (g == root) {
                for (int p = 0; p < G; ++p)
                    if (p != root) NCCLM(c, ncclSend(b + (size_t)p * per, per, ncclUint8, p, comms[l], streams[l]));
            } else NCCLM(c, ncclRecv(b + (size_t)g * per, per, ncclUint8, root, comms[l], streams[l]));
        }
        NCCLM(c, ncclGroupEnd());
        NCCLM(c, ncclGroupStart());
        for (int l = 0; l < nlocal(); ++l) {
            const int g = c->ranks[l].global;
            HIPM(c, hipSetDevice(c->ranks[l].device));
            char *b = static_cast<char *>(bufs[l]);
            NCCLM(c, ncclAllGather(b + (size_t)g * per, b, per, ncclUint8, comms[l], streams[l]));     // in place
        }
        NCCLM(c, ncclGroupEnd());
        return HJGPU_OK;
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
        NCCLM(c, ncclGroupStart());
        for (int l = 0; l < nlocal(); ++l) {
            HIPM(c, hipSetDevice(c->ranks[l].device));
            for (int p = 0; p < G; ++p) {
                const char *s = static_cast<const char *>(send[l]) + soff[l][p] * elem_bytes;
                for (u64 at = 0; at < scnt[l][p]; at += piece) {
                    const u64 n = scnt[l][p] - at < piece ? scnt[l][p] - at : piece;
                    NCCLM(c, ncclSend(s + at * elem_bytes, n * elem_bytes, ncclUint8, p, comms[l], streams[l]));
                }
                char *r = static_cast<char *>(recv[l]) + roff[l][p] * elem_bytes;
                for (u64 at = 0; at < rcnt[l][p]; at += piece) {
                    const u64 n = rcnt[l][p] - at < piece ? rcnt[l][p] - at : piece;
                    NCCLM(c, ncclRecv(r + at * elem_bytes, n * elem_bytes, ncclUint8, p, comms[l], streams[l]));
                }
            }
        }
        NCCLM(c, ncclGroupEnd());
        return HJGPU_OK;
    }

    int all_reduce_u64(u64 *const *bufs, size_t count, hipStream_t const *streams) override
    {
        NCCLM(c, ncclGroupStart());
        for (int l = 0; l < nlocal(); ++l) {
            HIPM(c, hipSetDevice(c->ranks[l].device));
            NCCLM(c, ncclAllReduce(bufs[l], bufs[l], count, ncclUint64, ncclSum, comms[l], streams[l]));
        }
        NCCLM(c, ncclGroupEnd());
        return HJGPU_OK;
    }
};

// ---- loopback: every rank in this process, messages are device-to-device copies -------------------------
__global__ void sum_rows_kernel(const u64 *__restrict__ rows, u64 *__restrict__ out, uint32_t nrows, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 s = 0;
    for (uint32_t r = 0; r < nrows; ++r) s += rows[(u64)r * count + i];
    out[i] = s;
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
    if (threadIdx.x < 4) dst[threadIdx.x] += src[threadIdx.x];
}

// ---- construction ---------------------------------------------------------------------------------
int init_rank(hjgpu_comm *c, Rank &r, int device, int global)
{
    r.device = device; r.global = global;
    int rc = hjgpu_create(device, &r.join);
    if (rc != HJGPU_OK) return cfail(c, rc, "hjgpu_create(join context)");
    rc = hjgpu_create(device, &r.part);
    if (rc != HJGPU_OK) return cfail(c, rc, "hjgpu_create(partition context)");
    HIPM(c, hipSetDevice(device));
    HIPM(c, hipStreamCreateWithFlags(&r.main, hipStreamNonBlocking));
    HIPM(c, hipStreamCreateWithFlags(&r.comm, hipStreamNonBlocking));
    HIPM(c, hipStreamCreateWithFlags(&r.prep, hipStreamNonBlocking));
    hipEvent_t *timed[] = {&r.ev_x0, &r.ev_x1, &r.ev_w0, &r.ev_w1};
    for (hipEvent_t *e : timed) HIPM(c, hipEventCreate(e));
    hipEvent_t *plain[] = {&r.ev_ready, &r.ev_part[0], &r.ev_part[1], &r.ev_xchg[0], &r.ev_xchg[1],
                           &r.ev_join[0], &r.ev_join[1], &r.ev_rx, &r.lb_in, &r.lb_out};
    for (hipEvent_t *e : plain) HIPM(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    const size_t G = (size_t)c->nranks;
    HIPM(c, hipHostMalloc(reinterpret_cast<void **>(&r.h_pin), (2 * (G + 1) + G * G + 8 + G) * sizeof(u64), hipHostMallocDefault));
    CHKM(ensure(c, r, r.d_off, 2 * (G + 1) * sizeof(u64)));
    CHKM(ensure(c, r, r.d_cnt, (G + G * G) * sizeof(u64)));
    CHKM(ensure(c, r, r.d_res, 8 * sizeof(u64)));
    return HJGPU_OK;
}

// RCCL's kernels need CUs WHILE the probe side is being partitioned (the build side arrives then; CPRA: slice i is on
// the links while slice i+1 is partitioned).  K6's workgroups fill a CU's LDS and live until their pass ends, so a
// kernel that arrives in the middle of a pass would wait for its end: the ranks' K6 grids leave some CUs free.
// That is free: K6 runs at the memory system's rate, not the CUs' - 64 M x 1 G on one GPU, ms per pass with
// 0 / 8 / 16 / 32 CUs left out: 3.01 / 3.03 / 3.03 / 3.13 and 3.04 / 3.03 / 3.03 / 3.06 (profiles/r02_reserve_sweep.txt).
int apply_reserve(hjgpu_comm *c)
{
    const bool rccl = c->transport && strcmp(c->transport->name(), "rccl") == 0;
    const int n = c->reserve_cus >= 0 ? c->reserve_cus : (rccl && c->nranks > 1 ? 16 : 0);
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
                   &r.d_res, &r.scratch};
    for (Buf *b : bufs) if (b->p) (void)hipFree(b->p);
    hipEvent_t evs[] = {r.ev_ready, r.ev_x0, r.ev_x1, r.ev_w0, r.ev_w1, r.ev_part[0], r.ev_part[1], r.ev_xchg[0],
                        r.ev_xchg[1], r.ev_join[0], r.ev_join[1], r.ev_rx, r.lb_in, r.lb_out};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    for (hipStream_t s : {r.main, r.comm, r.prep}) if (s) (void)hipStreamDestroy(s);
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

int sync_all(hjgpu_comm *c)
{
    for (Rank &r : c->ranks) {
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hipStreamSynchronize(r.prep));
        HIPM(c, hipStreamSynchronize(r.comm));
        HIPM(c, hipStreamSynchronize(r.main));
    }
    return HJGPU_OK;
}

void add_stats(hjgpu_stats *acc, const hjgpu_stats &s)
{
    acc->ms_total += s.ms_total; acc->ms_histogram += s.ms_histogram; acc->ms_plan += s.ms_plan;
    acc->ms_scatter1 += s.ms_scatter1; acc->ms_scatter2 += s.ms_scatter2; acc->ms_join += s.ms_join;
    acc->ms_build += s.ms_build; acc->ms_close_gaps += s.ms_close_gaps; acc->ms_inner_wait += s.ms_inner_wait;
    acc->fanout1 = s.fanout1; acc->fanout2 = s.fanout2; acc->buckets = s.buckets;
}

float elapsed(hipEvent_t a, hipEvent_t b)
{
    float ms = 0;
    return hipEventElapsedTime(&ms, a, b) == hipSuccess ? ms : 0.f;
}

// global sum of the ranks' running results (d_res[0..3]) -> host, on the join streams
int reduce_results(hjgpu_comm *c, hjgpu_result *result)
{
    std::vector<u64 *> res;
    for (Rank &r : c->ranks) res.push_back(static_cast<u64 *>(r.d_res.p));
    const std::vector<hipStream_t> mains = streams_of(c, &Rank::main);
    CHKM(c->transport->all_reduce_u64(res.data(), 4, mains.data()));
    const size_t G = (size_t)c->nranks;
    for (Rank &r : c->ranks) {
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hipMemcpyAsync(r.h_pin + 2 * (G + 1) + G * G, r.d_res.p, 4 * sizeof(u64), hipMemcpyDeviceToHost, r.main));
    }
    CHKM(sync_all(c));
    const u64 *h = c->ranks[0].h_pin + 2 * (G + 1) + G * G;
    if (result) { result->count = h[0]; result->sum_keys = h[1]; result->sum_outer_vals = h[2]; result->sum_inner_vals = h[3]; }
    return HJGPU_OK;
}

// ---- PHJ / NPJ: replicated build side, sharded probe side ---------------------------------------------
int replicated_join(hjgpu_comm *c, int algorithm, const hjgpu_shard *shards, int root,
                    const hjgpu_phj_params *pp, const hjgpu_npj_params *np, hjgpu_result *result,
                    hjgpu_multi_stats *stats)
{
    if (!c || !shards) return HJGPU_EINVAL;
    if (root < 0 || root >= c->nranks) return cfail(c, HJGPU_EINVAL, "root is not a rank of this communicator");
    const auto t0 = std::chrono::steady_clock::now();
    const int L = (int)c->ranks.size();
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
        HIPM(c, hipMemsetAsync(r.d_res.p, 0, 8 * sizeof(u64), r.main));
        HIPM(c, hipEventRecord(r.ev_x0, r.comm));
        if (r.global == root && inner) {
            uint32_t *b = static_cast<uint32_t *>(r.rbuf.p);
            HIPM(c, hipMemcpyAsync(b, shards[l].d_inner_keys, inner * sizeof(uint32_t), hipMemcpyDeviceToDevice, r.comm));
            HIPM(c, hipMemcpyAsync(b + stride, shards[l].d_inner_vals, inner * sizeof(uint32_t), hipMemcpyDeviceToDevice, r.comm));
        }
    }
    const std::vector<hipStream_t> comms = streams_of(c, &Rank::comm);
    if (inner) CHKM(c->transport->replicate(bufs.data(), bytes, root, c->ring_broadcast, comms.data()));
    for (int l = 0; l < L; ++l) {
        Rank &r = c->ranks[l];
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hipEventRecord(r.ev_x1, r.comm));
        HIPM(c, hipEventRecord(r.ev_ready, r.comm));
        const uint32_t *rk = static_cast<const uint32_t *>(r.rbuf.p), *rv = rk + stride;
        hjgpu_result *d_res = static_cast<hjgpu_result *>(r.d_res.p);
        if (algorithm == 1) {
            // the probe shard is histogrammed and partitioned while the build side is still arriving
            JOINM(c, r.join, hjgpu_phj_overlapped_async(r.join, rk, rv, inner, shards[l].d_outer_keys, shards[l].d_outer_vals,
                                                        shards[l].outer, pp, d_res, r.main, r.ev_ready));
        } else {
            HIPM(c, hipStreamWaitEvent(r.main, r.ev_ready, 0));     // NPJ builds first: it needs all of R
            JOINM(c, r.join, hjgpu_npj_async(r.join, rk, rv, inner, shards[l].d_outer_keys, shards[l].d_outer_vals,
                                             shards[l].outer, np, d_res, r.main));
        }
    }
    CHKM(reduce_results(c, result));
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
    return HJGPU_OK;
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
    bool exchange_in_flight = false;                          // local rank 0's ev_x0 / ev_x1 hold an unread exchange
    float exchange_ms = 0;
    CpraStep(hjgpu_comm *comm, hjgpu_multi_stats *st)
        : c(comm), L((int)comm->ranks.size()), G(comm->nranks), soff(L, std::vector<u64>(G)), scnt(L, std::vector<u64>(G)),
          roff(L, std::vector<u64>(G)), rcnt(L, std::vector<u64>(G)), recv_total(L), stats(st) {}

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
    int exchange(const std::vector<Slice> &in, int which, int slot)
    {
        const size_t Gs = (size_t)G;
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            ExchangeBufs b = bufs_of(r, which);
            CHKM(ensure(c, r, *b.sk, (in[l].n + 4) * sizeof(uint32_t)));
            CHKM(ensure(c, r, *b.sv, (in[l].n + 4) * sizeof(uint32_t)));
            u64 *d_off = static_cast<u64 *>(r.d_off.p) + (size_t)slot * (Gs + 1);
            u64 *h_off = r.h_pin + (size_t)slot * (Gs + 1);
            HIPM(c, hipSetDevice(r.device));
            if (in[l].n)
                JOINM(c, r.part, hjgpu_partition_async(r.part, in[l].keys, in[l].vals, in[l].n, TOP_LEVEL_FACTOR, (uint32_t)G,
                                                       static_cast<uint32_t *>(b.sk->p), static_cast<uint32_t *>(b.sv->p),
                                                       reinterpret_cast<uint64_t *>(d_off), r.prep));
            else HIPM(c, hipMemsetAsync(d_off, 0, (Gs + 1) * sizeof(u64), r.prep));
            HIPM(c, hipMemcpyAsync(h_off, d_off, (Gs + 1) * sizeof(u64), hipMemcpyDeviceToHost, r.prep));
            HIPM(c, hipEventRecord(r.ev_part[slot], r.prep));
        }
        // the host needs the counts: how much every peer gets decides the receive buffers
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            HIPM(c, hipSetDevice(r.device));
            HIPM(c, hipStreamSynchronize(r.prep));
            const u64 *h_off = r.h_pin + (size_t)slot * (Gs + 1);
            for (int p = 0; p < G; ++p) { soff[l][p] = h_off[p]; scnt[l][p] = h_off[p + 1] - h_off[p]; }
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
            u64 *h_cnt = r.h_pin + 2 * (Gs + 1) + Gs * Gs + 8;         // pinned: the copy is a DMA that runs later
            memcpy(h_cnt, scnt[l].data(), Gs * sizeof(u64));
            HIPM(c, hipMemcpyAsync(d_cnt, h_cnt, Gs * sizeof(u64), hipMemcpyHostToDevice, r.comm));
            csend.push_back(d_cnt); crecv.push_back(d_cnt + Gs);
        }
        const std::vector<hipStream_t> comms = streams_of(c, &Rank::comm);
        CHKM(c->transport->all_gather(csend.data(), crecv.data(), Gs * sizeof(u64), comms.data()));
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            HIPM(c, hipSetDevice(r.device));
            HIPM(c, hipMemcpyAsync(r.h_pin + 2 * (Gs + 1), static_cast<u64 *>(r.d_cnt.p) + Gs, Gs * Gs * sizeof(u64),
                                   hipMemcpyDeviceToHost, r.comm));
        }
        std::vector<const void *> ks, vs;
        std::vector<void *> kr, vr;
        std::vector<const u64 *> so, sc, ro, rc;
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            ExchangeBufs b = bufs_of(r, which);
            HIPM(c, hipSetDevice(r.device));
            HIPM(c, hipStreamSynchronize(r.comm));                  // also: the previous exchange has left the links
            if (l == 0) note_exchange();
            const u64 *matrix = r.h_pin + 2 * (Gs + 1);             // matrix[src][dst]
            u64 at = 0;
            for (int p = 0; p < G; ++p) { rcnt[l][p] = matrix[(size_t)p * Gs + r.global]; roff[l][p] = at; at += rcnt[l][p]; }
            recv_total[l] = at;
            // a quarter of headroom: the next slices rarely need a new allocation (hipFree waits for the device)
            if ((at + 4) * sizeof(uint32_t) > b.rk->cap) {
                CHKM(ensure(c, r, *b.rk, (at + at / 4 + 4) * sizeof(uint32_t)));
                CHKM(ensure(c, r, *b.rv, (at + at / 4 + 4) * sizeof(uint32_t)));
            }
            ks.push_back(b.sk->p); vs.push_back(b.sv->p); kr.push_back(b.rk->p); vr.push_back(b.rv->p);
            so.push_back(soff[l].data()); sc.push_back(scnt[l].data()); ro.push_back(roff[l].data()); rc.push_back(rcnt[l].data());
            HIPM(c, hipStreamWaitEvent(r.comm, r.ev_part[slot], 0));
            if (which) HIPM(c, hipStreamWaitEvent(r.comm, r.ev_join[slot], 0));   // the slot's previous slice has been joined
            if (l == 0) HIPM(c, hipEventRecord(r.ev_x0, r.comm));
        }
        CHKM(c->transport->all_to_all_v(ks.data(), so.data(), sc.data(), kr.data(), ro.data(), rc.data(), sizeof(uint32_t), comms.data()));
        CHKM(c->transport->all_to_all_v(vs.data(), so.data(), sc.data(), vr.data(), ro.data(), rc.data(), sizeof(uint32_t), comms.data()));
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            HIPM(c, hipSetDevice(r.device));
            if (l == 0) { HIPM(c, hipEventRecord(r.ev_x1, r.comm)); exchange_in_flight = true; }
            HIPM(c, hipEventRecord(bufs_of(r, which).done, r.comm));
        }
        return HJGPU_OK;
    }
};

int cpra_join(hjgpu_comm *c, const hjgpu_shard *shards, const hjgpu_phj_params *prm, int slices,
              hjgpu_result *result, hjgpu_multi_stats *stats)
{
    if (!c || !shards) return HJGPU_EINVAL;
    if (slices <= 0) slices = 4;
    if (slices > 4096) return cfail(c, HJGPU_EINVAL, "at most 4096 slices");
    const auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    const int L = (int)c->ranks.size();
    for (int l = 0; l < L; ++l) {
        const hjgpu_shard &s = shards[l];
        if ((s.inner && (!s.d_inner_keys || !s.d_inner_vals)) || (s.outer && (!s.d_outer_keys || !s.d_outer_vals)))
            return cfail(c, HJGPU_EINVAL, "null column in a shard");
    }
    CpraStep step(c, stats);
    for (Rank &r : c->ranks) {
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hipMemsetAsync(r.d_res.p, 0, 8 * sizeof(u64), r.main));
    }
    // ---- build side: partition the own chunk -> exchange -> prepared once for all probe slices ---------
    std::vector<Slice> in(L);
    for (int l = 0; l < L; ++l) in[l] = {shards[l].d_inner_keys, shards[l].d_inner_vals, shards[l].inner};
    CHKM(step.exchange(in, 0, 0));
    const std::vector<u64> inner_recv = step.recv_total;
    std::vector<size_t> max_outer(L);
    for (int l = 0; l < L; ++l) {
        Rank &r = c->ranks[l];
        // batches are the slices this rank RECEIVES: about one local slice when the hash spreads the keys evenly;
        // the workspace is sized for 1.5 of that, larger batches are probed in pieces
        const size_t per = shards[l].outer / (size_t)slices + 16;
        max_outer[l] = (per * 3 / 2 > ((size_t)1 << 20) ? per * 3 / 2 : ((size_t)1 << 20)) & ~size_t(15);
        HIPM(c, hipSetDevice(r.device));
        HIPM(c, hipStreamWaitEvent(r.main, r.ev_rx, 0));
        if (inner_recv[l])
            JOINM(c, r.join, hjgpu_phj_build(r.join, static_cast<const uint32_t *>(r.rrecv_k.p), static_cast<const uint32_t *>(r.rrecv_v.p),
                                             (size_t)inner_recv[l], max_outer[l], prm, r.main));
    }
    if (stats && inner_recv[0]) {
        // the build's phase times are read before the first probe re-records the context's events
        hjgpu_stats js;
        if (hjgpu_get_stats(c->ranks[0].join, &js) == HJGPU_OK) { add_stats(&stats->join, js); stats->joins += 1; stats->tuples_joined += inner_recv[0]; }
    }
    // ---- probe side in slices: partition(i+1) | exchange(i) | join(i-1) ----------------------------------
    // R join S = union_i (R join S_i): the slice results add up (add_result_kernel).
    auto join_slice = [&](int slot, const std::vector<u64> &got) -> int {
        u64 measured = 0;
        for (int l = 0; l < L; ++l) {
            Rank &r = c->ranks[l];
            HIPM(c, hipSetDevice(r.device));
            if (l == 0) HIPM(c, hipEventRecord(r.ev_w0, r.main));
            HIPM(c, hipStreamWaitEvent(r.main, r.ev_xchg[slot], 0));
            if (l == 0) HIPM(c, hipEventRecord(r.ev_w1, r.main));
            const uint32_t *sk = static_cast<const uint32_t *>(r.recv_k[slot].p), *sv = static_cast<const uint32_t *>(r.recv_v[slot].p);
            u64 *acc = static_cast<u64 *>(r.d_res.p);
            if (inner_recv[l])
                for (u64 b = 0; b < got[l]; b += max_outer[l]) {
                    const size_t m = got[l] - b < max_outer[l] ? (size_t)(got[l] - b) : max_outer[l];
                    JOINM(c, r.join, hjgpu_phj_probe_async(r.join, sk + b, sv + b, m, reinterpret_cast<hjgpu_result *>(acc + 4), r.main));
                    hipLaunchKernelGGL(add_result_kernel, dim3(1), dim3(64), 0, r.main, acc, acc + 4);
                    HIPM(c, hipGetLastError());
                    if (l == 0) measured = m;                        // the context's events describe its LAST batch
                }
            HIPM(c, hipEventRecord(r.ev_join[slot], r.main));
        }
        if (stats) {
            // read after every rank's work is enqueued (a single host thread must not wait in between)
            Rank &r = c->ranks[0];
            HIPM(c, hipSetDevice(r.device));
            hjgpu_stats js;
            if (measured && hjgpu_get_stats(r.join, &js) == HJGPU_OK) { add_stats(&stats->join, js); stats->joins += 1; stats->tuples_joined += measured; }
            HIPM(c, hipEventSynchronize(r.ev_w1));
            stats->ms_exchange_wait += elapsed(r.ev_w0, r.ev_w1);
        }
        return HJGPU_OK;
    };
    std::vector<u64> pending;
    int pending_slot = -1;
    for (int i = 0; i < slices; ++i) {
        const int slot = i & 1;
        for (int l = 0; l < L; ++l) {
            size_t b, e;
            range_of(shards[l].outer, 16, (size_t)i, (size_t)slices, &b, &e);
            in[l] = {shards[l].outer ? shards[l].d_outer_keys + b : nullptr, shards[l].outer ? shards[l].d_outer_vals + b : nullptr, e - b};
        }
        CHKM(step.exchange(in, 1 + slot, slot));
        if (pending_slot >= 0) CHKM(join_slice(pending_slot, pending));
        pending = step.recv_total;
        pending_slot = slot;
    }
    CHKM(join_slice(pending_slot, pending));
    CHKM(reduce_results(c, result));
    step.note_exchange();
    if (stats) {
        stats->ms_exchange = step.exchange_ms;
        stats->ms_wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return HJGPU_OK;
}

int new_comm(int nranks, hjgpu_comm **out, hjgpu_comm **c)
{
    if (!out) return HJGPU_EINVAL;
    *out = nullptr;
    if (nranks < 1 || nranks > 1024) return HJGPU_EINVAL;
    *c = new hjgpu_comm();
    (*c)->err[0] = 0;
    (*c)->nranks = nranks;
    return HJGPU_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

int hjgpu_comm_destroy(hjgpu_comm *c)
{
    if (!c) return HJGPU_OK;
    for (Rank &r : c->ranks) { (void)hipSetDevice(r.device); (void)hipDeviceSynchronize(); }
    c->transport.reset();                    // communicators before the streams they used
    for (Rank &r : c->ranks) destroy_rank(r);
    delete c;
    return HJGPU_OK;
}

int hjgpu_comm_create_local(int nranks, const int *devices, int transport, hjgpu_comm **out)
{
    hjgpu_comm *c = nullptr;
    CHKM(new_comm(nranks, out, &c));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete c; return HJGPU_ENODEVICE; }
    c->first = 0;
    c->ranks.resize((size_t)nranks);
    int rc = HJGPU_OK;
    std::vector<int> devs;
    for (int i = 0; i < nranks && rc == HJGPU_OK; ++i) {
        const int d = devices ? devices[i] : (transport == HJGPU_TRANSPORT_LOOPBACK ? i % ndev : i);
        if (d < 0 || d >= ndev) { rc = HJGPU_EINVAL; break; }
        for (int q : devs) if (q == d && transport == HJGPU_TRANSPORT_RCCL) rc = HJGPU_EINVAL;   // RCCL: one rank per device
        devs.push_back(d);
        if (rc == HJGPU_OK) rc = init_rank(c, c->ranks[(size_t)i], d, i);
    }
    if (rc == HJGPU_OK) {
        if (transport == HJGPU_TRANSPORT_LOOPBACK) c->transport.reset(new LoopbackTransport(c));
        else if (transport == HJGPU_TRANSPORT_RCCL) {
            RcclTransport *t = new RcclTransport(c);
            c->transport.reset(t);
            t->comms.assign((size_t)nranks, nullptr);
            const ncclResult_t r = ncclCommInitAll(t->comms.data(), nranks, devs.data());
            if (r != ncclSuccess) { fprintf(stderr, "hjgpu: ncclCommInitAll: %s\n", ncclGetErrorString(r)); rc = HJGPU_ERCCL; }
        } else rc = HJGPU_EINVAL;
    }
    if (rc == HJGPU_OK) rc = apply_reserve(c);
    if (rc != HJGPU_OK) { hjgpu_comm_destroy(c); return rc; }
    *out = c;
    return HJGPU_OK;
}

int hjgpu_comm_get_id(hjgpu_comm_id *id)
{
    static_assert(sizeof(ncclUniqueId) <= sizeof(hjgpu_comm_id), "hjgpu_comm_id too small for ncclUniqueId");
    if (!id) return HJGPU_EINVAL;
    memset(id, 0, sizeof(*id));
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return HJGPU_ERCCL;
    memcpy(id->bytes, &u, sizeof(u));
    return HJGPU_OK;
}

int hjgpu_comm_create_rank(int device, int nranks, int rank, const hjgpu_comm_id *id, hjgpu_comm **out)
{
    hjgpu_comm *c = nullptr;
    CHKM(new_comm(nranks, out, &c));
    if (!id || rank < 0 || rank >= nranks) { delete c; return HJGPU_EINVAL; }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) { delete c; return HJGPU_ENODEVICE; }
    c->first = rank;
    c->ranks.resize(1);
    int rc = init_rank(c, c->ranks[0], device, rank);
    if (rc == HJGPU_OK) {
        RcclTransport *t = new RcclTransport(c);
        c->transport.reset(t);
        t->comms.assign(1, nullptr);
        ncclUniqueId u;
        memcpy(&u, id->bytes, sizeof(u));
        if (hipSetDevice(device) != hipSuccess) rc = HJGPU_EHIP;
        else {
            const ncclResult_t r = ncclCommInitRank(&t->comms[0], nranks, u, rank);
            if (r != ncclSuccess) { fprintf(stderr, "hjgpu: ncclCommInitRank: %s\n", ncclGetErrorString(r)); rc = HJGPU_ERCCL; }
        }
    }
    if (rc == HJGPU_OK) rc = apply_reserve(c);
    if (rc != HJGPU_OK) { hjgpu_comm_destroy(c); return rc; }
    *out = c;
    return HJGPU_OK;
}

const char *hjgpu_comm_last_error(const hjgpu_comm *c) { return c ? c->err : "null communicator"; }

int hjgpu_comm_size(const hjgpu_comm *c, int *nranks, int *nlocal, int *first_rank)
{
    if (!c) return HJGPU_EINVAL;
    if (nranks) *nranks = c->nranks;
    if (nlocal) *nlocal = (int)c->ranks.size();
    if (first_rank) *first_rank = c->first;
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
    return cfail(c, HJGPU_EINVAL, "hjgpu_comm_set_option: unknown option");
}

int hjgpu_comm_barrier(hjgpu_comm *c)
{
    if (!c) return HJGPU_EINVAL;
    CHKM(sync_all(c));
    std::vector<u64 *> one;
    for (Rank &r : c->ranks) one.push_back(static_cast<u64 *>(r.d_cnt.p));
    const std::vector<hipStream_t> comms = streams_of(c, &Rank::comm);
    CHKM(c->transport->all_reduce_u64(one.data(), 1, comms.data()));
    return sync_all(c);
}

int hjgpu_phj_multi(hjgpu_comm *c, const hjgpu_shard *shards, int root, const hjgpu_phj_params *params,
                    hjgpu_result *result, hjgpu_multi_stats *stats)
{
    return replicated_join(c, 1, shards, root, params, nullptr, result, stats);
}

int hjgpu_npj_multi(hjgpu_comm *c, const hjgpu_shard *shards, int root, const hjgpu_npj_params *params,
                    hjgpu_result *result, hjgpu_multi_stats *stats)
{
    return replicated_join(c, 0, shards, root, nullptr, params, result, stats);
}

int hjgpu_cpra_multi(hjgpu_comm *c, const hjgpu_shard *shards, const hjgpu_phj_params *params, int slices,
                     hjgpu_result *result, hjgpu_multi_stats *stats)
{
    return cpra_join(c, shards, params, slices, result, stats);
}

int hjgpu_join_host_multi(hjgpu_comm *c, int algorithm,
                          const uint32_t *ik, const uint32_t *iv, size_t inner,
                          const uint32_t *ok, const uint32_t *ov, size_t outer,
                          const hjgpu_phj_params *pp, const hjgpu_npj_params *np,
                          hjgpu_result *result, hjgpu_multi_stats *stats)
{
    if (!c || algorithm < 0 || algorithm > 2) return HJGPU_EINVAL;
    if ((int)c->ranks.size() != c->nranks) return cfail(c, HJGPU_EINVAL, "hjgpu_join_host_multi needs a local communicator");
    if ((inner && (!ik || !iv)) || (outer && (!ok || !ov))) return cfail(c, HJGPU_EINVAL, "null column");
    const int G = c->nranks;
    std::vector<hjgpu_shard> shards((size_t)G);
    std::vector<void *> owned;
    int rc = HJGPU_OK;
    auto upload = [&](Rank &r, const uint32_t *h, size_t n, const uint32_t **d) {
        *d = nullptr;
        if (rc != HJGPU_OK) return;
        void *p = nullptr;
        if (hipSetDevice(r.device) != hipSuccess || hipMalloc(&p, (n + 4) * sizeof(uint32_t)) != hipSuccess) { rc = cfail(c, HJGPU_ENOMEM, "hipMalloc(shard)"); return; }
        owned.push_back(p);
        // pinned columns (hjgpu_host_alloc) are DMA'd; the GPUs' uploads then run side by side
        if (n && hipMemcpyAsync(p, h, n * sizeof(uint32_t), hipMemcpyHostToDevice, r.prep) != hipSuccess) { rc = cfail(c, HJGPU_EHIP, "hipMemcpyAsync(shard)"); return; }
        *d = static_cast<const uint32_t *>(p);
    };
    for (int g = 0; g < G; ++g) {
        Rank &r = c->ranks[(size_t)g];
        hjgpu_shard &s = shards[(size_t)g];
        memset(&s, 0, sizeof(s));
        size_t b, e;
        range_of(outer, 16, (size_t)g, (size_t)G, &b, &e);            // thread_beg / thread_end with T = ranks
        s.outer = e - b;
        upload(r, ok + b, s.outer, &s.d_outer_keys);
        upload(r, ov + b, s.outer, &s.d_outer_vals);
        if (algorithm == 2) {
            range_of(inner, 16, (size_t)g, (size_t)G, &b, &e);
            s.inner = e - b;
            upload(r, ik + b, s.inner, &s.d_inner_keys);
            upload(r, iv + b, s.inner, &s.d_inner_vals);
        } else {
            s.inner = inner;
            if (g == 0) { upload(r, ik, inner, &s.d_inner_keys); upload(r, iv, inner, &s.d_inner_vals); }
        }
    }
    if (rc == HJGPU_OK) rc = sync_all(c);
    if (rc == HJGPU_OK) {
        if (algorithm == 2) rc = cpra_join(c, shards.data(), pp, 0, result, stats);
        else rc = replicated_join(c, algorithm, shards.data(), 0, pp, np, result, stats);
    }
    (void)sync_all(c);
    for (void *p : owned) (void)hipFree(p);
    return rc;
}

}  // extern "C"
