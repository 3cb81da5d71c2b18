// cpp_pipeline_ordering.cpp - the PRODUCT's multi-GPU orchestration (csrc/hjgpu_multi.hip, included below as it is) on a CPU.
//
// The reference's workers cannot lose a tuple between two phases: they meet at barriers (cpra2.cpp:1834-1840,
// phj.cpp:1715-1770).  Here a phase is a kernel or a copy on one of a rank's streams and a barrier is a
// hipStreamWaitEvent / a host-side wait; whether every read of a buffer is ordered after the buffer's last write is a
// property of the ORDER in which hjgpu_cpra_multi / hjgpu_phj_multi / hjgpu_npj_multi enqueue things - checkable without
// a GPU.  This file supplies
//   * a HIP runtime that records (tests/mock_hip): every operation gets a vector clock over the streams (stream order,
//     event edges, host-side waits; host threads of each_rank through HJ_HOST_FORK / HJ_HOST_JOIN), names the memory it
//     reads and writes, and executes at once.  A read that is not ordered after the last write of what it reads, a write
//     not ordered after an earlier reader or writer: a VIOLATION.  Copies to host memory are delivered only when the
//     host waits for their stream: a host read without that wait sees stale counts and the join comes out wrong.
//   * the single-GPU library's entry points the orchestration calls (hjgpu_partition_packed*_async, hjgpu_phj_build_
//     prepartitioned, hjgpu_phj_probe_prepartitioned*_async, hjgpu_phj_overlapped_async, hjgpu_npj_async, ...) as plain
//     CPU code with the same contracts - partitions are really made, pieces really checked (a tuple in a piece of a rank
//     that does not own its partition, a sender's count that differs from what arrived: an ERROR), joins really joined -
//     so the result of every scenario is compared with a map-based join of the inputs;
//   * fault injection: --drop-wait s<stream>#<n> ignores the n-th hipStreamWaitEvent ON THAT STREAM (the checker must then report).
// usage: cpp_pipeline_ordering <algo cpra|cpra-host|phj-host|npj-host|phj|npj> <world> <slices> [--rows] [--no-fused] [--no-in-place] [--two-level] [--grouped]
//                              [--drop-wait s<stream>#<n>] [--list-waits] [--inner n] [--outer n] [--seed s] [--steps n]
// prints one line: "ok|FAIL waits=<hipStreamWaitEvent calls> ops=<n> violations=<n> ..."; exit status 0 = result right and no violation.
#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <unordered_map>

// ---- recorder state (declared before the product file: HJ_HOST_FORK / HJ_HOST_JOIN) -----------------------------------
namespace rec {
void host_fork();
void host_join();
}
#define HJ_HOST_FORK() rec::host_fork()
#define HJ_HOST_JOIN() rec::host_join()

#include "../hash_join_codes_knl_amd/csrc/hjgpu_multi.hip"

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace rec {

typedef std::vector<uint32_t> Clock;                      // [stream id] = operations of that stream known to have finished

std::recursive_mutex mu;
struct Op { int stream; uint32_t index; std::string what; };
std::vector<Op> ops;
int violations = 0, errors = 0;
std::vector<std::string> messages;
long wait_calls = 0;
// a wait is named by the stream that waits and its ordinal among THAT stream's waits ("s5#3"): every stream is fed by one host thread at a time
// (each_rank: one thread per rank), so the name means the same wait in every run - a global counter over all ranks' threads did not
// (round 5: the wait dropped by --drop-wait k was not always wait k of the --list-waits run)
int drop_stream = -1; long drop_ordinal = -1;
bool list_waits = false;
std::vector<std::string> wait_list;          // --list-waits: every hipStreamWaitEvent: which stream, the stream the event was recorded on, did it add an edge

void complain(bool error, const std::string &m)
{
    (error ? errors : violations) += 1;
    if (messages.size() < 12) messages.push_back(m);
}

void merge(Clock &a, const Clock &b)
{
    if (a.size() < b.size()) a.resize(b.size(), 0);
    for (size_t i = 0; i < b.size(); ++i) a[i] = std::max(a[i], b[i]);
}

// host threads: what each knows to have finished
thread_local Clock host_clock;
thread_local bool host_known = false;
Clock fork_clock, join_clock;
Clock &host()
{
    if (!host_known) { host_clock = fork_clock; host_known = true; }     // a thread made by each_rank knows what its maker knew
    return host_clock;
}
void host_fork() { std::lock_guard<std::recursive_mutex> g(mu); fork_clock = host(); join_clock.clear(); }
void host_join() { std::lock_guard<std::recursive_mutex> g(mu); merge(host(), join_clock); }
void host_learned() { merge(join_clock, host()); }          // (under mu) a worker's knowledge reaches its maker at the join

}  // namespace rec

struct MockStream {
    int id;
    uint32_t issued = 0;                 // operations enqueued so far
    long waits = 0;                      // hipStreamWaitEvent calls on this stream so far
    rec::Clock clock;                    // what the NEXT operation of this stream is ordered after
    struct Pending { void *dst; std::vector<unsigned char> data; };
    std::vector<Pending> to_host;        // copies to host memory, delivered when the host waits for the stream
};
struct MockEvent { rec::Clock clock; bool recorded = false; int source = -1; };

namespace rec {

std::vector<MockStream *> streams;
// memory: allocation base -> size; per 8-byte word the last writer and the readers since
struct Access { int stream; uint32_t index; int op; };
struct Word { Access w{-1, 0, -1}; std::vector<Access> r; };
struct Block { size_t bytes; bool host; std::vector<Word> words; std::string name; };
std::map<uintptr_t, Block> blocks;
int next_block = 0;

Block *find(const void *p, uintptr_t *base)
{
    auto it = blocks.upper_bound((uintptr_t)p);
    if (it == blocks.begin()) return nullptr;
    --it;
    if ((uintptr_t)p >= it->first + it->second.bytes) return nullptr;
    *base = it->first;
    return &it->second;
}

bool before(const Access &a, const Clock &c) { return a.stream < 0 || ((size_t)a.stream < c.size() && c[(size_t)a.stream] >= a.index); }

// one operation on `s`: its clock, then its reads and writes
struct Scope {
    MockStream *s;
    Clock clock;
    int op;
    Scope(MockStream *st, const std::string &what) : s(st)
    {
        merge(s->clock, host());                                  // enqueued after everything this host thread has waited for
        s->issued += 1;
        if (s->clock.size() <= (size_t)s->id) s->clock.resize((size_t)s->id + 1, 0);
        s->clock[(size_t)s->id] = s->issued;
        clock = s->clock;
        op = (int)ops.size();
        ops.push_back({s->id, s->issued, what});
    }
    void touch(const void *p, size_t bytes, bool write)
    {
        if (!bytes || !p) return;
        uintptr_t base = 0;
        Block *b = find(p, &base);
        if (!b) return;                                           // not memory the runtime made (a stack variable of the caller)
        const size_t first = ((uintptr_t)p - base) / 8, last = std::min(b->words.size(), ((uintptr_t)p - base + bytes + 7) / 8);
        for (size_t i = first; i < last; ++i) {
            Word &w = b->words[i];
            if (!before(w.w, clock) && !(w.w.stream == s->id))
                complain(false, "op " + std::to_string(op) + " (" + ops[(size_t)op].what + ", stream " + std::to_string(s->id) + ") " + (write ? "writes" : "reads") + " " + b->name + "+" +
                                    std::to_string(i * 8) + " unordered after its write by op " + std::to_string(w.w.op) + " (" + ops[(size_t)w.w.op].what + ", stream " + std::to_string(w.w.stream) + ")");
            if (write) {
                for (const Access &r : w.r)
                    if (!before(r, clock) && r.stream != s->id)
                        complain(false, "op " + std::to_string(op) + " (" + ops[(size_t)op].what + ", stream " + std::to_string(s->id) + ") overwrites " + b->name + "+" + std::to_string(i * 8) +
                                            " unordered after its read by op " + std::to_string(r.op) + " (" + ops[(size_t)r.op].what + ", stream " + std::to_string(r.stream) + ")");
                w.r.clear();
                w.w = {s->id, s->issued, op};
            } else {
                bool seen = false;
                for (Access &r : w.r) if (r.stream == s->id) { r.index = s->issued; r.op = op; seen = true; }
                if (!seen) w.r.push_back({s->id, s->issued, op});
            }
        }
    }
    void reads(const void *p, size_t bytes) { touch(p, bytes, false); }
    void writes(void *p, size_t bytes) { touch(p, bytes, true); }
};

void host_waited(MockStream *s)
{
    for (MockStream::Pending &p : s->to_host) memcpy(p.dst, p.data.data(), p.data.size());
    s->to_host.clear();
    Clock done = s->clock;
    if (done.size() <= (size_t)s->id) done.resize((size_t)s->id + 1, 0);
    done[(size_t)s->id] = s->issued;
    merge(host(), done);
    host_learned();
}

MockStream null_stream{0};

MockStream *of(hipStream_t s) { return s ? s : streams[0]; }

}  // namespace rec

// ---- the HIP runtime ----------------------------------------------------------------------------------------------
#define LOCK std::lock_guard<std::recursive_mutex> guard_(rec::mu)
static void *new_block(size_t bytes, bool host, const char *kind)
{
    void *p = calloc(1, bytes + 64);
    rec::Block b;
    b.bytes = bytes + 64; b.host = host; b.words.resize((bytes + 64 + 7) / 8);
    b.name = std::string(kind) + std::to_string(rec::next_block++) + "[" + std::to_string(bytes) + "]";
    rec::blocks[(uintptr_t)p] = std::move(b);
    return p;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 8; return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipDeviceSynchronize() { LOCK; for (MockStream *s : rec::streams) rec::host_waited(s); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    LOCK;
    if (rec::streams.empty()) { rec::streams.push_back(new MockStream{0}); }
    *s = new MockStream{(int)rec::streams.size()};
    rec::streams.push_back(*s);
    return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { LOCK; rec::host_waited(rec::of(s)); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { LOCK; rec::host_waited(rec::of(s)); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    LOCK;
    rec::wait_calls++;
    MockStream *ws = rec::of(s);
    const long k = ws->waits++;
    if (rec::list_waits) {
        // an edge is NEW when the waiting stream (with what the enqueuing host thread has waited for) does not know the event's clock yet
        rec::Clock known = rec::of(s)->clock;
        rec::merge(known, rec::host());
        bool fresh = false;
        for (size_t i = 0; e->recorded && i < e->clock.size(); ++i) if (e->clock[i] > (i < known.size() ? known[i] : 0u)) fresh = true;
        rec::wait_list.push_back("s" + std::to_string(ws->id) + "#" + std::to_string(k) + ":s" + std::to_string(ws->id) + "<-s" + std::to_string(e->source) + (fresh ? ":new" : ":known"));
    }
    if (ws->id == rec::drop_stream && k == rec::drop_ordinal) return hipSuccess;       // fault injection: this wait never happened
    if (e->recorded) rec::merge(rec::of(s)->clock, e->clock);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new MockEvent; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s_)
{
    LOCK;
    MockStream *s = rec::of(s_);
    rec::merge(s->clock, rec::host());
    e->clock = s->clock;
    if (e->clock.size() <= (size_t)s->id) e->clock.resize((size_t)s->id + 1, 0);
    e->clock[(size_t)s->id] = s->issued;
    e->recorded = true;
    e->source = s->id;
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { LOCK; if (e->recorded) { rec::merge(rec::host(), e->clock); rec::host_learned(); } return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes) { LOCK; *p = new_block(bytes, false, "dev"); return hipSuccess; }
hipError_t hipFree(void *p)
{
    LOCK;
    if (!p) return hipSuccess;
    for (MockStream *s : rec::streams) rec::host_waited(s);        // hipFree waits for the device
    rec::blocks.erase((uintptr_t)p);
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { LOCK; *p = new_block(bytes, true, "pinned"); return hipSuccess; }
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t s_)
{
    LOCK;
    MockStream *s = rec::of(s_);
    rec::Scope op(s, "memcpy " + std::to_string(bytes));
    op.reads(src, bytes); op.writes(dst, bytes);
    uintptr_t base = 0;
    rec::Block *b = rec::find(dst, &base);
    if (b && b->host) {                                           // to host memory: there when the host has waited for the stream
        MockStream::Pending p{dst, std::vector<unsigned char>((const unsigned char *)src, (const unsigned char *)src + bytes)};
        s->to_host.push_back(std::move(p));
    } else if (bytes) memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind k)
{
    LOCK;
    hipMemcpyAsync(dst, src, bytes, k, nullptr);
    rec::host_waited(rec::of(nullptr));
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t s_)
{
    LOCK;
    rec::Scope op(rec::of(s_), "memset " + std::to_string(bytes));
    op.writes(dst, bytes);
    memset(dst, v, bytes);
    return hipSuccess;
}
// the library's own clears and device-to-device copies (csrc/gen_kernels.hip: kernels with non-temporal stores): recorded like the runtime's
hipError_t hj_zero_async(void *p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s); }
hipError_t hj_copy_async(void *dst, const void *src, size_t bytes, hipStream_t s) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s); }
hipError_t hipMemset(void *dst, int v, size_t bytes) { LOCK; hipMemsetAsync(dst, v, bytes, nullptr); rec::host_waited(rec::of(nullptr)); return hipSuccess; }
hipError_t hipLaunchHostFunc(hipStream_t s_, hipHostFn_t, void *) { LOCK; rec::Scope op(rec::of(s_), "host function"); return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "mock hip error"; }

void mock_note_kernel(const char *name, hipStream_t s_, const std::vector<const void *> &p, const std::vector<uint64_t> &i)
{
    LOCK;
    rec::Scope op(rec::of(s_), name);
    const std::string n = name;
    if (n == "sum_rows_kernel") { op.reads(p[0], (size_t)i[0] * i[1] * 8); op.writes(const_cast<void *>(p[1]), (size_t)i[1] * 8); }
    else if (n == "add_result_kernel") { op.reads(p[1], 32); op.reads(p[0], 32); op.writes(const_cast<void *>(p[0]), 32); }
    else if (n == "bump_kernel") { op.reads(p[0], 8); op.writes(const_cast<void *>(p[0]), 8); }
    else rec::complain(true, "a kernel the recorder does not know: " + n);
}

// ---- the single-GPU library, as far as the orchestration calls it ------------------------------------------------------
static inline uint32_t H(uint32_t key, uint32_t f, uint32_t n) { return (uint32_t)(((uint64_t)(uint32_t)(key * f) * n) >> 32); }

struct hjgpu_ctx {
    int id;
    char err[256];
    void *ws_build, *ws_probe;                       // stand-ins for the workspace: the prepared build side / the per-call scratch
    std::unordered_multimap<uint32_t, uint32_t> build;
    bool prepared = false, pre = false;
    hjgpu_prepartitioned lay;
    uint32_t F2 = 0, f2 = 0;
    hjgpu_output out;
    bool has_out = false;
    uint64_t flags[2] = {0, 0};
};
static int n_ctx = 0;
static int cfail_ctx(hjgpu_ctx *c, int st, const char *m) { snprintf(c->err, sizeof(c->err), "%s", m); return st; }

int hjgpu_create(int, hjgpu_ctx **out)
{
    LOCK;
    hjgpu_ctx *c = new hjgpu_ctx;
    c->id = n_ctx++; c->err[0] = 0;
    c->ws_build = new_block(64, false, "build-workspace"); c->ws_probe = new_block(64, false, "call-workspace");
    *out = c;
    return HJGPU_OK;
}
int hjgpu_destroy(hjgpu_ctx *) { return HJGPU_OK; }
const char *hjgpu_last_error(const hjgpu_ctx *c) { return c ? c->err : "null"; }
const char *hjgpu_status_string(int) { return "status"; }
int hjgpu_set_option(hjgpu_ctx *, const char *, const char *) { return HJGPU_OK; }
int hjgpu_get_stats(hjgpu_ctx *, hjgpu_stats *s) { memset(s, 0, sizeof(*s)); return HJGPU_OK; }
int hjgpu_audit_read(hjgpu_ctx *, uint64_t *next, uint64_t, uint32_t, uint64_t *, void *) { if (next) *next = 0; return HJGPU_OK; }
int hjgpu_audit_recheck(hjgpu_ctx *, uint64_t *, size_t, size_t *checks) { if (checks) *checks = 0; return HJGPU_OK; }
int hjgpu_malloc_placed(hjgpu_ctx *, void **p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess ? HJGPU_OK : HJGPU_ENOMEM; }
int hjgpu_host_alloc(hjgpu_ctx *, void **p, size_t bytes) { return hipHostMalloc(p, bytes, 0) == hipSuccess ? HJGPU_OK : HJGPU_ENOMEM; }
int hjgpu_set_async_output(hjgpu_ctx *c, const hjgpu_output *o) { LOCK; c->out = *o; c->has_out = true; return HJGPU_OK; }
int hjgpu_output_capacity(hjgpu_ctx *, int, size_t, size_t rows, size_t bs, size_t *cap) { *cap = (rows / (bs ? bs : 65536) + 2) * (bs ? bs : 65536); return HJGPU_OK; }
int hjgpu_accumulate_async_status(hjgpu_ctx *c, uint64_t *d_flags, void *stream)
{
    LOCK;
    rec::Scope op(rec::of((hipStream_t)stream), "accumulate_async_status");
    op.reads(c->ws_probe, 8); op.reads(d_flags, 16); op.writes(d_flags, 16);
    d_flags[0] += c->flags[0]; d_flags[1] += c->flags[1];
    return HJGPU_OK;
}

// exchange-level partitioning: packed tuples, plain prefix of the counts in d_offsets, own partitions last
static int partition_packed(hjgpu_ctx *c, const uint32_t *k, const uint32_t *v, size_t n, uint32_t factor, uint32_t F, uint32_t own_first,
                            uint32_t own_count, uint32_t f2, uint32_t F2, uint64_t *out, uint64_t *off, uint64_t *counts2, void *stream,
                            const char *what)
{
    LOCK;
    rec::Scope op(rec::of((hipStream_t)stream), what);
    op.reads(k, n * 4); op.reads(v, n * 4); op.reads(c->ws_probe, 8); op.writes(c->ws_probe, 8);
    std::vector<uint64_t> cnt(F, 0), prefix(F + 1, 0);
    for (size_t i = 0; i < n; ++i) cnt[H(k[i], factor, F)] += 1;
    for (uint32_t p = 0; p < F; ++p) prefix[p + 1] = prefix[p] + cnt[p];
    const uint64_t own_rows = prefix[own_first + own_count] - prefix[own_first];
    std::vector<uint64_t> at(F);
    for (uint32_t p = 0; p < F; ++p) {
        at[p] = prefix[p];
        if (p >= own_first + own_count) at[p] -= own_rows;
        else if (p >= own_first) at[p] = n - own_rows + (prefix[p] - prefix[own_first]);
    }
    op.writes(out, n * 8); op.writes(off, ((size_t)F + 1) * 8);
    if (counts2) { op.writes(counts2, (size_t)F * F2 * 8); memset(counts2, 0, (size_t)F * F2 * 8); }
    for (size_t i = 0; i < n; ++i) {
        const uint32_t p = H(k[i], factor, F);
        out[at[p]++] = ((uint64_t)v[i] << 32) | k[i];
        if (counts2) counts2[(size_t)p * F2 + H(k[i], f2, F2)] += 1;
    }
    memcpy(off, prefix.data(), ((size_t)F + 1) * 8);
    return HJGPU_OK;
}
int hjgpu_partition_packed_async(hjgpu_ctx *c, const uint32_t *k, const uint32_t *v, size_t n, uint32_t f, uint32_t F, uint64_t *out, uint64_t *off, void *s)
{ return partition_packed(c, k, v, n, f, F, 0, 0, 0, 0, out, off, nullptr, s, "partition_packed"); }
int hjgpu_partition_packed_own_last_async(hjgpu_ctx *c, const uint32_t *k, const uint32_t *v, size_t n, uint32_t f, uint32_t F, uint32_t of, uint32_t oc,
                                          uint64_t *out, uint64_t *off, void *s)
{ return partition_packed(c, k, v, n, f, F, of, oc, 0, 0, out, off, nullptr, s, "partition_packed_own_last"); }
int hjgpu_partition_packed_counted_async(hjgpu_ctx *c, const uint32_t *k, const uint32_t *v, size_t n, uint32_t f, uint32_t F, uint32_t of, uint32_t oc,
                                         uint32_t f2, uint32_t F2, uint64_t *out, uint64_t *off, uint64_t *counts2, void *s)
{ return partition_packed(c, k, v, n, f, F, of, oc, f2, F2, out, off, counts2, s, "partition_packed_counted"); }
int hjgpu_partition_async(hjgpu_ctx *c, const uint32_t *k, const uint32_t *v, size_t n, uint32_t factor, uint32_t F, uint32_t *ok, uint32_t *ov,
                          uint64_t *off, void *stream)
{
    LOCK;
    rec::Scope op(rec::of((hipStream_t)stream), "partition (columns)");
    op.reads(k, n * 4); op.reads(v, n * 4); op.reads(c->ws_probe, 8); op.writes(c->ws_probe, 8);
    std::vector<uint64_t> prefix(F + 1, 0);
    for (size_t i = 0; i < n; ++i) prefix[H(k[i], factor, F) + 1] += 1;
    for (uint32_t p = 0; p < F; ++p) prefix[p + 1] += prefix[p];
    std::vector<uint64_t> at(prefix.begin(), prefix.end() - 1);
    op.writes(ok, n * 4); op.writes(ov, n * 4); op.writes(off, ((size_t)F + 1) * 8);
    for (size_t i = 0; i < n; ++i) { const uint64_t d = at[H(k[i], factor, F)]++; ok[d] = k[i]; ov[d] = v[i]; }
    memcpy(off, prefix.data(), ((size_t)F + 1) * 8);
    return HJGPU_OK;
}
bool mock_grouped = false;                   // --grouped: the planning rule says "grouped" (the ranks then take CPRA's grouped road)
int hjgpu_grouped_plan(hjgpu_ctx *, size_t, size_t, const hjgpu_phj_params *, uint32_t *groups) { *groups = mock_grouped ? 4 : 0; return HJGPU_OK; }
// (a device-planned grouped local join is enqueue-only; the orchestration waits for its stream - with its deadline - and asks for the status)
int hjgpu_get_async_status(hjgpu_ctx *, void *s) { hipStreamSynchronize((hipStream_t)s); return HJGPU_OK; }
int hjgpu_prepartitioned_plan(hjgpu_ctx *, size_t, uint32_t, const hjgpu_phj_params *prm, uint32_t *F2, uint32_t *f2)
{
    *F2 = (prm && prm->fanout2) ? prm->fanout2 : 5;
    *f2 = (prm && prm->factor2) ? prm->factor2 : 0x85EBCA6Bu;
    return HJGPU_OK;
}

static void join_rows(hjgpu_ctx *c, rec::Scope &op, uint32_t key, uint32_t val, uint64_t r[4], uint64_t *rows_written)
{
    auto range = c->build.equal_range(key);
    for (auto it = range.first; it != range.second; ++it) {
        r[0] += 1; r[1] += key; r[2] += val; r[3] += it->second;
        if (c->has_out) {
            if (*rows_written < c->out.capacity) {
                c->out.d_keys[*rows_written] = key; c->out.d_outer_vals[*rows_written] = val; c->out.d_inner_vals[*rows_written] = it->second;
            } else c->flags[1] = 1;
            *rows_written += 1;
        }
    }
    (void)op;
}
static void finish_join(hjgpu_ctx *c, rec::Scope &op, const uint64_t r[4], uint64_t rows_written, hjgpu_result *d_result)
{
    if (c->has_out) {
        const uint64_t n = std::min<uint64_t>(rows_written, c->out.capacity);
        op.writes(c->out.d_keys, n * 4); op.writes(c->out.d_outer_vals, n * 4); op.writes(c->out.d_inner_vals, n * 4);
        c->has_out = false;
    }
    if (d_result) { op.writes(d_result, 32); d_result->count = r[0]; d_result->sum_keys = r[1]; d_result->sum_outer_vals = r[2]; d_result->sum_inner_vals = r[3]; }
}

int hjgpu_phj_build_prepartitioned(hjgpu_ctx *c, const uint64_t *t, const hjgpu_prepartitioned *lay, size_t, const hjgpu_phj_params *prm, void *stream)
{
    LOCK;
    rec::Scope op(rec::of((hipStream_t)stream), "build_prepartitioned");
    const uint64_t b = lay->chunk_offsets[0], e = lay->chunk_offsets[lay->chunks];
    op.reads(t + b, (e - b) * 8); op.writes(c->ws_build, 8); op.writes(c->ws_probe, 8);
    c->build.clear();
    for (uint64_t i = b; i < e; ++i) {
        const uint32_t key = (uint32_t)t[i], p = H(key, lay->factor1, lay->fanout1_total);
        if (p < lay->first_partition || p >= lay->first_partition + lay->fanout1) rec::complain(true, "build side: a tuple arrived at a rank that does not own its partition");
        c->build.emplace(key, (uint32_t)(t[i] >> 32));
    }
    c->lay = *lay; c->pre = true; c->prepared = true; c->flags[0] = c->flags[1] = 0;
    hjgpu_prepartitioned_plan(c, 0, lay->fanout1, prm, &c->F2, &c->f2);
    return HJGPU_OK;
}
static int probe_pre(hjgpu_ctx *c, const uint64_t *t, const hjgpu_prepartitioned *lay, const uint64_t *counts, hjgpu_result *d_result, void *stream)
{
    LOCK;
    rec::Scope op(rec::of((hipStream_t)stream), counts ? "probe_prepartitioned_counted" : "probe_prepartitioned");
    if (!c->prepared || !c->pre) return cfail_ctx(c, HJGPU_EINVAL, "no prepared build side");
    const uint64_t b = lay->chunk_offsets[0], e = lay->chunk_offsets[lay->chunks];
    op.reads(t + b, (e - b) * 8); op.reads(c->ws_build, 8); op.reads(c->ws_probe, 8); op.writes(c->ws_probe, 8);
    const size_t P = (size_t)lay->fanout1 * c->F2;
    std::vector<uint64_t> seen((size_t)lay->chunks * P, 0);
    uint64_t r[4] = {0, 0, 0, 0}, rows = 0;
    for (uint32_t piece = 0; piece < lay->chunks; ++piece)
        for (uint64_t i = lay->chunk_offsets[piece]; i < lay->chunk_offsets[piece + 1]; ++i) {
            const uint32_t key = (uint32_t)t[i], p = H(key, lay->factor1, lay->fanout1_total);
            if (p < lay->first_partition || p >= lay->first_partition + lay->fanout1) { rec::complain(true, "probe side: a tuple arrived at a rank that does not own its partition"); continue; }
            seen[(size_t)piece * P + (size_t)(p - lay->first_partition) * c->F2 + H(key, c->f2, c->F2)] += 1;
            join_rows(c, op, key, (uint32_t)(t[i] >> 32), r, &rows);
        }
    if (counts) {
        op.reads(counts, seen.size() * 8);
        if (memcmp(counts, seen.data(), seen.size() * 8) != 0) rec::complain(true, "the senders' counts differ from what arrived (fused counts)");
    }
    finish_join(c, op, r, rows, d_result);
    return HJGPU_OK;
}
int hjgpu_phj_probe_prepartitioned_async(hjgpu_ctx *c, const uint64_t *t, const hjgpu_prepartitioned *lay, hjgpu_result *r, void *s) { return probe_pre(c, t, lay, nullptr, r, s); }
int hjgpu_phj_probe_prepartitioned_counted_async(hjgpu_ctx *c, const uint64_t *t, const hjgpu_prepartitioned *lay, const uint64_t *cn, hjgpu_result *r, void *s)
{ return probe_pre(c, t, lay, cn, r, s); }

int hjgpu_phj_build(hjgpu_ctx *c, const uint32_t *k, const uint32_t *v, size_t n, size_t, const hjgpu_phj_params *, void *stream)
{
    LOCK;
    rec::Scope op(rec::of((hipStream_t)stream), "build (columns)");
    op.reads(k, n * 4); op.reads(v, n * 4); op.writes(c->ws_build, 8); op.writes(c->ws_probe, 8);
    c->build.clear();
    for (size_t i = 0; i < n; ++i) c->build.emplace(k[i], v[i]);
    c->pre = false; c->prepared = true; c->flags[0] = c->flags[1] = 0;
    return HJGPU_OK;
}
int hjgpu_phj_probe_async(hjgpu_ctx *c, const uint32_t *k, const uint32_t *v, size_t n, hjgpu_result *d_result, void *stream)
{
    LOCK;
    rec::Scope op(rec::of((hipStream_t)stream), "probe (columns)");
    if (!c->prepared || c->pre) return cfail_ctx(c, HJGPU_EINVAL, "no prepared build side");
    op.reads(k, n * 4); op.reads(v, n * 4); op.reads(c->ws_build, 8); op.reads(c->ws_probe, 8); op.writes(c->ws_probe, 8);
    uint64_t r[4] = {0, 0, 0, 0}, rows = 0;
    for (size_t i = 0; i < n; ++i) join_rows(c, op, k[i], v[i], r, &rows);
    finish_join(c, op, r, rows, d_result);
    return HJGPU_OK;
}
// the probe side is partitioned first; the stream waits for `inner_ready` right before the first read of the build side
int hjgpu_phj_overlapped_async(hjgpu_ctx *c, const uint32_t *rk, const uint32_t *rv, size_t inner, const uint32_t *sk, const uint32_t *sv, size_t outer,
                               const hjgpu_phj_params *, hjgpu_result *d_result, void *stream, void *inner_ready)
{
    {
        LOCK;
        rec::Scope op(rec::of((hipStream_t)stream), "overlapped: probe side partitioned");
        op.reads(sk, outer * 4); op.reads(sv, outer * 4); op.reads(c->ws_probe, 8); op.writes(c->ws_probe, 8);
    }
    if (inner_ready) hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)inner_ready, 0);
    LOCK;
    rec::Scope op(rec::of((hipStream_t)stream), "overlapped: build side + join");
    op.reads(rk, inner * 4); op.reads(rv, inner * 4); op.reads(c->ws_probe, 8); op.writes(c->ws_probe, 8); op.writes(c->ws_build, 8);
    c->build.clear();
    for (size_t i = 0; i < inner; ++i) c->build.emplace(rk[i], rv[i]);
    c->flags[0] = c->flags[1] = 0;
    uint64_t r[4] = {0, 0, 0, 0}, rows = 0;
    for (size_t i = 0; i < outer; ++i) join_rows(c, op, sk[i], sv[i], r, &rows);
    finish_join(c, op, r, rows, d_result);
    return HJGPU_OK;
}
int hjgpu_npj_async(hjgpu_ctx *c, const uint32_t *rk, const uint32_t *rv, size_t inner, const uint32_t *sk, const uint32_t *sv, size_t outer,
                    const hjgpu_npj_params *, hjgpu_result *d_result, void *stream)
{
    LOCK;
    rec::Scope op(rec::of((hipStream_t)stream), "npj");
    op.reads(rk, inner * 4); op.reads(rv, inner * 4); op.reads(sk, outer * 4); op.reads(sv, outer * 4);
    op.reads(c->ws_probe, 8); op.writes(c->ws_probe, 8); op.writes(c->ws_build, 8);
    c->build.clear();
    c->flags[0] = c->flags[1] = 0;
    for (size_t i = 0; i < inner; ++i) { if (rk[i] == 0) c->flags[0] = 1; else c->build.emplace(rk[i], rv[i]); }
    uint64_t r[4] = {0, 0, 0, 0}, rows = 0;
    for (size_t i = 0; i < outer; ++i) join_rows(c, op, sk[i], sv[i], r, &rows);
    finish_join(c, op, r, rows, d_result);
    return HJGPU_OK;
}
// the host-column entry points are not part of this test (they need the single-GPU host pipeline)
int hjgpu_join_host(hjgpu_ctx *c, int, const uint32_t *, const uint32_t *, size_t, const uint32_t *, const uint32_t *, size_t, const hjgpu_phj_params *,
                    const hjgpu_npj_params *, hjgpu_result *, hjgpu_stats *) { return cfail_ctx(c, HJGPU_EINVAL, "not in the mock"); }
int hjgpu_join_host_rows_shared(hjgpu_ctx *c, int, const uint32_t *, const uint32_t *, size_t, const uint32_t *, const uint32_t *, size_t,
                                const hjgpu_phj_params *, const hjgpu_npj_params *, const hjgpu_host_rows *, uint64_t *, hjgpu_result *, hjgpu_stats *)
{ return cfail_ctx(c, HJGPU_EINVAL, "not in the mock"); }

// ---- scenarios ------------------------------------------------------------------------------------------------------
static uint64_t rng_state = 1;
static uint32_t rnd()
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s cpra|phj|npj world slices [options]\n", argv[0]); return 2; }
    const std::string algo = argv[1];
    const int G = atoi(argv[2]), slices = atoi(argv[3]);
    bool rows = false, fused = true, in_place = true, two_level = false;
    size_t inner = 3000, outer = 20000;
    int steps = 2;
    for (int i = 4; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--rows") rows = true;
        else if (a == "--no-fused") fused = false;
        else if (a == "--no-in-place") in_place = false;
        else if (a == "--two-level") two_level = true;
        else if (a == "--grouped") mock_grouped = true;
        else if (a == "--drop-wait" && i + 1 < argc) { if (sscanf(argv[++i], "s%d#%ld", &rec::drop_stream, &rec::drop_ordinal) != 2) { fprintf(stderr, "--drop-wait s<stream>#<ordinal>\n"); return 2; } }
        else if (a == "--list-waits") rec::list_waits = true;
        else if (a == "--inner" && i + 1 < argc) inner = (size_t)atol(argv[++i]);
        else if (a == "--outer" && i + 1 < argc) outer = (size_t)atol(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) rng_state = (uint64_t)atol(argv[++i]);
        else if (a == "--steps" && i + 1 < argc) steps = atoi(argv[++i]);
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    std::vector<int> devices((size_t)G, 0);
    hjgpu_comm *comm = nullptr;
    if (hjgpu_comm_create_local(G, devices.data(), HJGPU_TRANSPORT_LOOPBACK, &comm) != HJGPU_OK) { fprintf(stderr, "create: %s\n", hjgpu_comm_last_error(nullptr)); return 2; }
    hjgpu_comm_set_option(comm, "cpra_fused_counts", fused ? "1" : "0");
    hjgpu_comm_set_option(comm, "exchange_in_place", in_place ? "1" : "0");
    hjgpu_comm_set_option(comm, "cpra_two_level", two_level ? "1" : "0");
    hjgpu_comm_set_option(comm, "cpra_grouped", mock_grouped ? "2" : "1");      // 2: the road whenever the planning rule groups
    // relations: unique non-zero build keys, probe keys drawn from them and from outside, ragged shares (one rank may get nothing)
    std::vector<uint32_t> ik(inner), iv(inner), ok(outer), ov(outer);
    for (size_t i = 0; i < inner; ++i) { ik[i] = (uint32_t)(i + 1) * 2654435761u | 1u; iv[i] = rnd(); }
    std::sort(ik.begin(), ik.end());
    ik.erase(std::unique(ik.begin(), ik.end()), ik.end());
    inner = ik.size();
    for (size_t i = inner; i > 1; --i) std::swap(ik[i - 1], ik[rnd() % i]);
    for (size_t i = 0; i < outer; ++i) { ok[i] = (rnd() % 8) ? ik[rnd() % inner] : rnd() | 1u; ov[i] = rnd(); }
    std::unordered_map<uint32_t, uint32_t> truth;
    for (size_t i = 0; i < inner; ++i) truth[ik[i]] = iv[i];
    uint64_t want[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < outer; ++i) {
        auto it = truth.find(ok[i]);
        if (it != truth.end()) { want[0] += 1; want[1] += ok[i]; want[2] += ov[i]; want[3] += it->second; }
    }
    // cut points: multiples of 16 rows (the alignment every entry point asks for), ragged
    auto cuts = [&](size_t n) {
        std::vector<size_t> c((size_t)G + 1, 0);
        for (int g = 1; g < G; ++g) c[(size_t)g] = std::min(n, ((n * (size_t)g / (size_t)G + (rnd() % 64) * 16) & ~size_t(15)));
        c[(size_t)G] = n;
        std::sort(c.begin(), c.end());
        if (G > 2) c[2] = c[1];                                  // one rank holds nothing
        std::sort(c.begin(), c.end());
        return c;
    };
    if (algo == "cpra-host" || algo == "phj-host" || algo == "npj-host") {
        const int algorithm = algo == "cpra-host" ? 2 : algo == "phj-host" ? 1 : 0;
        if (rows && algorithm != 2) { fprintf(stderr, "rows from host columns with a replicated build side run the one-GPU host pipeline per rank: not in this test\n"); return 2; }
        // hjgpu_join_host_multi / hjgpu_join_host_rows_multi, CPRA: host columns cut into the ranks' shares, the build side uploaded
        // first, the probe shard slice by slice (an upload event per slice), the slice pipeline behind it
        bool right = true;
        for (int step = 0; step < steps; ++step) {
            hjgpu_result got;
            memset(&got, 0, sizeof(got));
            std::vector<uint32_t> hk(outer + 1024), ho(outer + 1024), hi(outer + 1024);
            hjgpu_host_rows hr = {hk.data(), ho.data(), hi.data(), hk.size()};
            const int rc = rows ? hjgpu_join_host_rows_multi(comm, algorithm, ik.data(), iv.data(), inner, ok.data(), ov.data(), outer, nullptr, nullptr, &hr, &got, nullptr)
                                : hjgpu_join_host_multi(comm, algorithm, ik.data(), iv.data(), inner, ok.data(), ov.data(), outer, nullptr, nullptr, &got, nullptr);
            if (rc != HJGPU_OK) { fprintf(stderr, "step %d: status %d: %s\n", step, rc, hjgpu_comm_last_error(comm)); right = false; break; }
            if (got.count != want[0] || got.sum_keys != want[1] || got.sum_outer_vals != want[2] || got.sum_inner_vals != want[3]) right = false;
            if (rows) {
                uint64_t sk = 0;
                for (uint64_t i = 0; i < got.count && i < hk.size(); ++i) {
                    auto it = truth.find(hk[i]);
                    if (it == truth.end() || it->second != hi[i]) { right = false; break; }
                    sk += hk[i];
                }
                if (sk != want[1]) right = false;
            }
        }
        const bool ok_all = right && rec::violations == 0 && rec::errors == 0;
        printf("%s waits=%ld ops=%zu violations=%d errors=%d result=%s\n", ok_all ? "ok" : "FAIL", rec::wait_calls, rec::ops.size(), rec::violations, rec::errors,
               right ? "right" : "WRONG");
        for (const std::string &m : rec::messages) printf("  %s\n", m.c_str());
        if (rec::list_waits) { printf("waits:"); for (const std::string &w : rec::wait_list) printf(" %s", w.c_str()); printf("\n"); }
        return ok_all ? 0 : 1;
    }
    const bool replicated = algo != "cpra";
    const std::vector<size_t> ci = cuts(inner), co = cuts(outer);
    std::vector<hjgpu_shard> shards((size_t)G);
    std::vector<hjgpu_shard_rows> srows((size_t)G);
    std::vector<std::vector<uint32_t *>> cols((size_t)G);
    auto dev = [&](const uint32_t *src, size_t n) {
        uint32_t *p = nullptr;
        hipMalloc(reinterpret_cast<void **>(&p), (n + 16) * 4);
        if (n && src) memcpy(p, src, n * 4);
        return p;
    };
    for (int g = 0; g < G; ++g) {
        hjgpu_shard &s = shards[(size_t)g];
        memset(&s, 0, sizeof(s));
        const size_t ib = replicated ? 0 : ci[(size_t)g], ie = replicated ? inner : ci[(size_t)g + 1];
        s.inner = ie - ib; s.outer = co[(size_t)g + 1] - co[(size_t)g];
        if (!replicated || g == G - 1) { s.d_inner_keys = dev(ik.data() + ib, s.inner); s.d_inner_vals = dev(iv.data() + ib, s.inner); }
        s.d_outer_keys = dev(ok.data() + co[(size_t)g], s.outer); s.d_outer_vals = dev(ov.data() + co[(size_t)g], s.outer);
        hjgpu_shard_rows &r = srows[(size_t)g];
        memset(&r, 0, sizeof(r));
        r.out.block_size = 256; r.out.capacity = ((outer + 255) / 256 + 2) * 256;
        r.out.d_keys = dev(nullptr, r.out.capacity); r.out.d_outer_vals = dev(nullptr, r.out.capacity); r.out.d_inner_vals = dev(nullptr, r.out.capacity);
    }
    bool right = true;
    for (int step = 0; step < steps; ++step) {                   // twice: the second step meets the first one's buffers and events
        hjgpu_result got;
        memset(&got, 0, sizeof(got));
        int rc;
        if (algo == "cpra") rc = rows ? hjgpu_cpra_multi_rows(comm, shards.data(), srows.data(), nullptr, slices, &got, nullptr) : hjgpu_cpra_multi(comm, shards.data(), nullptr, slices, &got, nullptr);
        else if (algo == "phj") rc = rows ? hjgpu_phj_multi_rows(comm, shards.data(), srows.data(), G - 1, nullptr, &got, nullptr) : hjgpu_phj_multi(comm, shards.data(), G - 1, nullptr, &got, nullptr);
        else rc = rows ? hjgpu_npj_multi_rows(comm, shards.data(), srows.data(), G - 1, nullptr, &got, nullptr) : hjgpu_npj_multi(comm, shards.data(), G - 1, nullptr, &got, nullptr);
        if (rc != HJGPU_OK) { fprintf(stderr, "step %d: status %d: %s\n", step, rc, hjgpu_comm_last_error(comm)); right = false; break; }
        if (got.count != want[0] || got.sum_keys != want[1] || got.sum_outer_vals != want[2] || got.sum_inner_vals != want[3]) {
            fprintf(stderr, "step %d: count %llu (want %llu)\n", step, (unsigned long long)got.count, (unsigned long long)want[0]);
            right = false;
        }
        if (rows) {
            // the ranks' rows together are the result: every row a match, as many as counted
            uint64_t total = 0, sk = 0;
            for (int g = 0; g < G; ++g) {
                total += srows[(size_t)g].rows;
                for (uint64_t i = 0; i < srows[(size_t)g].rows; ++i) {
                    auto it = truth.find(srows[(size_t)g].out.d_keys[i]);
                    if (it == truth.end() || it->second != srows[(size_t)g].out.d_inner_vals[i]) { right = false; break; }
                    sk += srows[(size_t)g].out.d_keys[i];
                }
            }
            if (total != want[0] || sk != want[1]) { fprintf(stderr, "step %d: rows %llu (want %llu)\n", step, (unsigned long long)total, (unsigned long long)want[0]); right = false; }
        }
    }
    const bool ok_all = right && rec::violations == 0 && rec::errors == 0;
    printf("%s waits=%ld ops=%zu violations=%d errors=%d result=%s\n", ok_all ? "ok" : "FAIL", rec::wait_calls, rec::ops.size(), rec::violations, rec::errors,
           right ? "right" : "WRONG");
    for (const std::string &m : rec::messages) printf("  %s\n", m.c_str());
    if (rec::list_waits) { printf("waits:"); for (const std::string &w : rec::wait_list) printf(" %s", w.c_str()); printf("\n"); }
    return ok_all ? 0 : 1;
}
