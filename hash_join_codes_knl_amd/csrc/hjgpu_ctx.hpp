// hjgpu_ctx.hpp - what the translation units behind include/hjgpu.h share: the context, its workspace layout, the plan of a
// PHJ / CPRA join, and the planning / enqueue functions of hjgpu_api.hip that the operator-level entry points (hjgpu_ops.hip)
// and the host pipelines (hjgpu_host.hip) call.  Internal: nothing outside csrc/ includes it.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

#include "hj_internal.hpp"


struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// Small device-resident state of one join.
struct DevState {
    hjgpu_result result;
    u64 block_counter;
    u64 dense;
    u64 work_counter;
    uint32_t overflow;
    uint32_t zero_key;
    uint32_t nmoves;
    uint32_t pad;
    u64 work_counter2;      // the multi-fill half of a _UNIQUE join (hj_launch_join)
    uint32_t group_skew;    // a device-planned grouped join skipped a group that was larger than its workspace (group_desc_kernel)
    uint32_t pad2;
};

// probe side (S) is partitioned first, then the build side (R): a caller can overlap the
// arrival of R (e.g. an RCCL broadcast) with the S passes through `inner_ready`
enum { EV_BEGIN = 0, EV_S_HIST, EV_S_PLAN, EV_S_SC1, EV_S_SC2, EV_WAITED,
       EV_R_HIST, EV_R_PLAN, EV_R_SC1, EV_R_SC2, EV_JOIN, EV_GAPS, EV_COUNT };


struct hjgpu_ctx {
    int device = 0;
    int cus = 0;
    hipDeviceProp_t prop;
    char err[512];
    DevBuf tmp[8];          // pass-1 / pass-2 twins of the 4 columns (hj.h's [1] scratch columns)
    DevBuf meta;            // histograms, offsets, cursors, tile / work-item prefixes
    DevBuf table;           // NPJ table
    DevBuf state;           // DevState
    DevBuf moves;           // close_gaps move list
    DevBuf final_offsets;   // per-wave end cursors
    hipEvent_t ev[EV_COUNT];
    bool ev_valid[EV_COUNT];
    hjgpu_stats stats;
    int last_algo = -1;     // 0 npj, 1 phj/cpra
    // hjgpu_phj_build: the partitioned build side (tmp[0] / tmp[4]) and its plan (meta) stay valid until
    // another entry point uses the workspace
    bool prepared = false;
    size_t prepared_inner = 0, prepared_max_outer = 0;
    unsigned char prepared_plan[128];
    HjTuning tune;          // tuning / test switches: environment at hjgpu_create, hjgpu_set_option afterwards
    // hjgpu_join_host*: two page-locked staging buffers for PAGEABLE host columns, made when the first one is seen and kept
    // (hipHostMalloc + hipHostFree of 2 x 32 MiB cost 22 ms per call; page-locked columns never need them)
    // the batched host calls' three streams (upload / join / download), made once: the runtime binds a stream's copies to a
    // DMA engine when it first uses it, and fresh streams in every call ended up with upload and download on ONE engine
    // from the second call on (one after the other: 430 ms instead of 275 for 8.5 GB up and 12 GB down)
    hipStream_t host_streams[3] = {nullptr, nullptr, nullptr};
    void *host_stage[4] = {nullptr, nullptr, nullptr, nullptr};          // [0..1] uploads, [2..3] downloads (both run at once
    hipEvent_t host_stage_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // when result rows go home behind the upload)
    // hjgpu_set_async_output: the next *_async join of this context materialises into these columns (one-shot)
    hjgpu_output pending_out;
    bool has_pending_out = false;
    bool last_had_output = false;   // hjgpu_get_async_status: the last enqueued join wrote result columns
    bool rows_plain = false;        // the join being enqueued is SOLO - a blocking call of a context with option "solo": plain partial-line stores
                                    // in K6 (k6_store8) and plain result rows (join_kernel<..., NTROWS = false>); every other launch writes them non-temporal
    // grouped plans: pass-0 twins of the four columns, the groups' offsets and (device-planned) descriptors, and the call's accumulated
    // phase times (hjgpu_get_stats returns those while stats_override is set; any later operation's first event clears it)
    DevBuf grp[4], grp_off;
    bool stats_override = false;
    hipStream_t aux = nullptr;      // private non-blocking stream: placement probes of the workspace allocator
    // Device-planned grouped joins (phj_grouped_device): one set of phase events per group - the groups' joins are enqueued back to back
    // and nobody waits in between, so hjgpu_get_stats adds the groups' spans up afterwards - and what hjgpu_get_async_status needs to
    // run the join again, host-planned, when a group turned out larger than its workspace (DevState::group_skew)
    hipEvent_t *ev_cur = nullptr;                        // the set record() writes (NULL: ev)
    std::vector<hipEvent_t> grp_ev;                      // [groups][EV_COUNT]
    uint32_t grp_ev_groups = 0;                          // groups of the last device-planned join (0: hjgpu_get_stats reads ev)
    hipEvent_t grp_ev_pass0[3] = {nullptr, nullptr, nullptr};   // begin, pass 0 done, all done
    struct GroupedCall {
        bool valid = false;
        uint32_t chunks = 1;
        const uint32_t *rk = nullptr, *rv = nullptr, *sk = nullptr, *sv = nullptr;
        size_t inner = 0, outer = 0;
        bool has_prm = false, has_out = false;
        hjgpu_phj_params prm;
        hjgpu_output out;
        hjgpu_result *d_result = nullptr;
    } grp_last;                                          // the last ENQUEUE-ONLY device-planned grouped join of this context
    // option "audit": the last HJ_AUDIT_RING calls' stage records (audit_kernels.hip), the next call's sequence number, and the
    // explicit partition bounds of an own-last layout
    DevBuf audit, audit_lay;
    uint64_t audit_seq = 0;
    // the partition checks of the LAST audited call, as they were launched: hjgpu_audit_recheck does them again with the device quiet
    struct AuditCheck { int stage; const u64 *tuples, *beg, *end; uint32_t parts; HjAuditHash h; };
    std::vector<AuditCheck> audit_checks;
    float ms_reserve = 0;           // wall clock of the workspace growth so far (allocations + placement probes)
    // the last placement search (ensure_placed): candidate blocks it allocated and filled, the kept block's fill time and size,
    // whether the budget (option "placement_ms") ended it
    uint32_t placement_tried = 0, placement_timeboxed = 0;
    float placement_fill_ms = 0, placement_search_ms = 0;
    size_t placement_bytes = 0;
};

namespace hjapi {

const uint32_t DEFAULT_F1 = 0x9E3779B1u, DEFAULT_F2 = 0x85EBCA6Bu;
const uint32_t DEFAULT_TF0 = 0xC2B2AE35u, DEFAULT_TF1 = 0x27D4EB2Fu;
const uint32_t DEFAULT_NPJ_FACTOR = 0x9E3779B1u;
const uint32_t DEFAULT_F0 = 0x7FEB352Du;      // grouped plans: pass 0 (phj_grouped takes another one when a pass factor of the join equals it)

int fail(hjgpu_ctx *ctx, int status, const char *what, hipError_t e = hipSuccess);

// The partial-line stores (K6) of a SOLO join - a blocking call of a context with option "solo": the caller waits for it and promises
// that nothing else runs on the device beside it - are plain; every other join (enqueue-only, host batches, multi-GPU slices, and
// every blocking call without the option) writes them non-temporal (DESIGN section 3 "Round 5").  The same for the result rows of PHJ / CPRA
// (join_kernel's NTROWS instances); NPJ's rows are always non-temporal.
struct PlainRows {
    hjgpu_ctx *ctx;
    PlainRows(hjgpu_ctx *c, bool blocking) : ctx(c) { if (ctx) ctx->rows_plain = blocking && ctx->tune.solo; }
    ~PlainRows() { if (ctx) ctx->rows_plain = false; }
};

#define HIPCHK(ctx, call)                                                       \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) return fail((ctx), HJGPU_EHIP, #call, e_);        \
    } while (0)
hipError_t hj_event_synchronize(hipEvent_t ev);
hipError_t hj_stream_synchronize(hipStream_t st);

#define CHK(call)                                                               \
    do {                                                                        \
        int s_ = (call);                                                        \
        if (s_ != HJGPU_OK) return s_;                                          \
    } while (0)

int ensure(hjgpu_ctx *ctx, DevBuf &b, size_t bytes);
int ensure_placed(hjgpu_ctx *ctx, DevBuf &b, size_t bytes);

// Wall clock of workspace growth (hjgpu_stats.ms_reserve): the placement search holds and fills up to 12 candidate
// blocks of the probe side's pass-1 twin; it happens at hjgpu_reserve / the first join of a size, never in a timed join
// afterwards, and its cost is reported instead of being invisible.
struct ReserveClock {
    hjgpu_ctx *ctx;
    size_t before;
    std::chrono::steady_clock::time_point t0;
    static size_t held(const hjgpu_ctx *c)
    {
        size_t n = c->meta.cap + c->table.cap + c->state.cap;
        for (const DevBuf &b : c->tmp) n += b.cap;
        return n;
    }
    explicit ReserveClock(hjgpu_ctx *c) : ctx(c), before(held(c)), t0(std::chrono::steady_clock::now()) {}
    ~ReserveClock()
    {
        if (held(ctx) != before)
            ctx->ms_reserve += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};

uint32_t grouped_groups(const hjgpu_ctx *ctx, size_t inner, size_t outer, const hjgpu_phj_params *prm);
struct GroupLayout { uint32_t G, bins, F0; };
GroupLayout group_layout(uint32_t G);
int grouped_twins(hjgpu_ctx *ctx, const GroupLayout &l, size_t inner, size_t outer);

// K6 occupies a CU completely (one 1024-thread workgroup with ~155 KiB of LDS that lives until the pass ends): a kernel
// that arrives during a pass - RCCL's, on the multi-GPU path - finds no CU until the pass is over.  "reserve_cus" keeps
// some CUs out of K6's grid (work is claimed from a ticket counter, so any grid size finishes the pass).
inline int scatter_cus(const hjgpu_ctx *ctx)
{
    const int n = ctx->cus - ctx->tune.reserve_cus;
    return n < 1 ? 1 : n;
}

inline uint32_t align_of(const void *p) { return (uint32_t)(((uintptr_t)p >> 2) & 3); }

// Carves the meta buffer; must match between sizing and use.
struct MetaLayout {
    u64 *counts[2], *off2[2], *end2[2], *cur2[2], *off1[2], *cur1[2], *tp1[2], *seg1[2], *tp2[2];
    u64 *seg2[2];                // [F1 + 1] partition-major pass-1 layout of a chunked relation: bounds of the pass-1 partitions
    u64 *more[2];                // [P] more than 8 chunks: the counters of chunks 8 ... C - 1 added up (PlanArgs::more), else NULL
    u64 *slice_prefix, *slices;
    uint32_t *tickets;           // [HJ_TICKET_WORDS] work-claim counters of K4 / K6, the multi-fill count (inside the block zeroed per join)
    uint32_t *item_part;         // [P + items_extra] partition of every join work item
    uint4 *tdesc[2];             // [tdesc_cap][2] pass-2 tile descriptors (K5 -> K6 pass 2)
    size_t tdesc_cap;
    uint32_t *range_counts[2];   // [ranges][F1] pass-1 counts per range (K4 -> K5b)
    u64 *range_base[2];          // [ranges][F1] pass-1 write bases per range (K5b -> K6)
    // batched probe-side partitioning (hj_launch_batch_plan): per-batch pass-1 layout, pass-2 tile prefix,
    // pass-2 tile descriptors, and one ticket word per launch (pass 1 / pass 2 of every batch)
    u64 *boff, *tp2b;
    uint4 *tdescb;
    uint32_t *btickets;
    size_t btickets_bytes;
    size_t counts_bytes;    // both relations, contiguous (zeroed per join)
    size_t total_bytes;
};
MetaLayout carve(void *base, uint32_t C, uint32_t F1, uint32_t P, size_t ranges, size_t items_extra = 0,
                 size_t tiles2 = 0, size_t batches = 0, size_t tdesc_b_cap = 0);
void choose_fanout(const HjTuning &tune, size_t inner, const hjgpu_phj_params *prm, uint32_t *F1, uint32_t *F2, bool *big_tables);
void record(hjgpu_ctx *ctx, int which, hipStream_t s);
int audit_begin(hjgpu_ctx *ctx, int kind, size_t inner, size_t outer, hipStream_t stream, u64 **rec);
// hj_audit_partitions into stage `stage` of the call's record, remembered for hjgpu_audit_recheck
int audit_partitions(hjgpu_ctx *ctx, int stage, const u64 *tuples, const u64 *beg, const u64 *end, uint32_t parts, const HjAuditHash &h, u64 *rec,
                     hipStream_t stream);
int refuse_capture(hjgpu_ctx *ctx, hipStream_t stream);
int check_columns(hjgpu_ctx *ctx, const uint32_t *k, const uint32_t *v, size_t n);
int setup_output(hjgpu_ctx *ctx, const hjgpu_output *out, uint32_t workers, u64 *block_size, u64 *block_limit);
uint32_t range_tiles_for(const HjTuning &tune, u64 max_tiles, uint32_t F1);
uint32_t ranges_of(const HjTuning &tune, u64 max_tiles, uint32_t F1);
uint32_t ranges_capacity(const HjTuning &tune, u64 max_tiles, uint32_t F1);
Pass1Geom make_geom(const HjTuning &tune, const void *keys, size_t n, uint32_t C, uint32_t F1, bool out_packed, bool capacity = false);

struct PhjPlan {
    size_t ranges, items_extra, tiles2;
    uint32_t C, F1, F2, P;
    uint32_t f1, f2, tf0, tf1;
    bool big_tables;
    bool unique;             // HJGPU_FLAG_UNIQUE / option "unique"
    // batched probe-side partitioning: 0 batches = off
    uint32_t batch_ranges;   // pass-1 ranges per batch
    uint32_t batch_cap;      // batches the tables hold
    uint32_t batch_tile_cap; // tiles per range the batch buffers are sized for
    size_t tdesc_b_cap;      // pass-2 tile descriptors per batch
    size_t batch_bytes;      // one batch buffer (packed tuples)
    // pre-partitioned relations (hjgpu_phj_build_prepartitioned): pass 1 was the exchange-level partitioning of the
    // multi-GPU CPRA; F1 = this rank's share k of its fan-out pre_F1tot, partitions [pre_base, pre_base + k)
    uint32_t pre;            // 1: the relations arrive pass-1-partitioned
    uint32_t pre_f1, pre_F1tot, pre_base;
};
static_assert(sizeof(PhjPlan) <= sizeof(hjgpu_ctx::prepared_plan), "prepared_plan too small");
// the pieces a pre-partitioned relation arrives in (one per source rank)
struct PrePieces {
    const u64 *tuples[2] = {nullptr, nullptr};      // [0] build side, [1] probe side (packed: payload << 32 | key)
    HjChunks ch[2];
    // [chunks][P] fused (piece, final partition) counts of the relation, counted by the SENDERS' histogram pass and
    // delivered with the exchange (hjgpu_phj_probe_prepartitioned_counted_async): K4p is skipped
    const u64 *counts[2] = {nullptr, nullptr};
};
enum PhjMode { PHJ_WHOLE = 0, PHJ_BUILD_ONLY = 1, PHJ_PROBE_ONLY = 2 };

// plan_inner > 0: the fan-out is planned for a build side of that many rows while the workspace holds `inner` (device-planned groups: the
// mean group decides the partitions, the largest group the plan allows decides the buffers)
int phj_prepare(hjgpu_ctx *ctx, size_t inner, size_t outer, const hjgpu_phj_params *prm, uint32_t chunks, PhjPlan *pl, bool pre = false,
                int big_override = -1, size_t plan_inner = 0);
// One group of a device-planned grouped join: desc = {build first row, build rows, probe first row, probe rows} in device memory; the
// columns handed to phj_enqueue are the pass-0 twins, `inner` / `outer` the CAPACITY the plan was prepared for.  The join's state
// (aggregates, block counter, open output blocks) is the whole grouped join's: phj_enqueue neither clears it nor closes the gaps.
struct GroupRun { const u64 *desc; };
int phj_enqueue(hjgpu_ctx *ctx, const PhjPlan &pl, const uint32_t *rk, const uint32_t *rv, size_t inner,
                const uint32_t *sk, const uint32_t *sv, size_t outer, const hjgpu_output *out, hipStream_t stream,
                hipEvent_t inner_ready = nullptr, PhjMode mode = PHJ_WHOLE, const PrePieces *pre = nullptr, const GroupRun *grp = nullptr);
int finish_blocking(hjgpu_ctx *ctx, hjgpu_result *result, const hjgpu_output *out, hipStream_t stream);
bool npj_unique(const hjgpu_ctx *ctx, const hjgpu_npj_params *prm);
int npj_prepare(hjgpu_ctx *ctx, size_t inner, const hjgpu_npj_params *prm, size_t *buckets, uint32_t *factor);
int npj_probe_enqueue(hjgpu_ctx *ctx, const uint32_t *sk, const uint32_t *sv, size_t outer, const u64 *table, size_t buckets, uint32_t factor,
                      const hjgpu_output *out, hipStream_t stream, bool line_hash = false, bool unique = false);
int npj_enqueue(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner, const uint32_t *sk, const uint32_t *sv, size_t outer,
                size_t buckets, uint32_t factor, const hjgpu_output *out, hipStream_t stream, bool unique);
const hjgpu_output *take_async_output(hjgpu_ctx *ctx, const hjgpu_output *given);
// hjgpu_ops.hip: the partition operator on separate columns (hjgpu_partition_async; pass 0 of a grouped plan: group_bins > 0)
int partition_columns(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n, uint32_t factor, uint32_t fanout,
                      uint32_t group_bins, uint32_t *d_keys_out, uint32_t *d_vals_out, uint64_t *d_offsets, void *stream_);

}  // namespace hjapi
