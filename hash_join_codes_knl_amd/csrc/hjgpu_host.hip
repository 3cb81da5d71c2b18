// hjgpu_host.hip - the host-column entry points of include/hjgpu.h: hjgpu_join_host / _rows / _rows_shared.  The reference's
// mains fread the four columns and then start their clock (npj.cpp:1013-1039, 861); here the columns cross PCIe first: pinned
// uploads, the probe side in batches behind the DMA, a batch's rows on their way home while the next batch is joined.
#include "hjgpu_ctx.hpp"

using namespace hjapi;

extern "C" {

// Host column -> HBM on `copy`.  Page-locked memory (hjgpu_host_alloc, hipHostRegister'ed, ...) is
// DMA'd directly; pageable memory goes through two pinned staging buffers so that the CPU's copy
// of chunk i+1 overlaps the DMA of chunk i.
constexpr size_t HJ_HOST_STAGE = 32u << 20;
static int host_stage(hjgpu_ctx *ctx, int first)
{
    for (int b = first; b < first + 2; ++b) {
        if (!ctx->host_stage[b]) HIPCHK(ctx, hipHostMalloc(&ctx->host_stage[b], HJ_HOST_STAGE, hipHostMallocDefault));
        if (!ctx->host_stage_ev[b]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->host_stage_ev[b], hipEventDisableTiming));
    }
    return HJGPU_OK;
}

static int upload_column(hjgpu_ctx *ctx, void *d, const void *h, size_t bytes, hipStream_t copy, int *next)
{
    if (!bytes) return HJGPU_OK;
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, h) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();                            // a pageable pointer reports an error: expected
    if (pinned) {
        HIPCHK(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, copy));
        return HJGPU_OK;
    }
    CHK(host_stage(ctx, 0));
    void **stage = ctx->host_stage;
    hipEvent_t *stage_free = ctx->host_stage_ev;
    const size_t stage_bytes = HJ_HOST_STAGE;
    for (size_t at_ = 0; at_ < bytes; at_ += stage_bytes) {
        const size_t n = bytes - at_ < stage_bytes ? bytes - at_ : stage_bytes;
        const int b = *next; *next ^= 1;
        HIPCHK(ctx, hj_event_synchronize(stage_free[b]));           // the DMA that last used this buffer is done
        memcpy(stage[b], (const char *)h + at_, n);
        HIPCHK(ctx, hipMemcpyAsync((char *)d + at_, stage[b], n, hipMemcpyHostToDevice, copy));
        HIPCHK(ctx, hipEventRecord(stage_free[b], copy));
    }
    return HJGPU_OK;
}

// HBM column -> host column on `copy`: the mirror image of upload_column.  Pageable destinations are
// filled from two pinned staging buffers, the CPU's copy of chunk i overlapping the DMA of chunk i+1.
static int download_column(hjgpu_ctx *ctx, void *h, const void *d, size_t bytes, hipStream_t copy, bool by_kernel = false)
{
    if (!bytes) return HJGPU_OK;
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, h) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    // (by_kernel: the DMA engines are busy with an upload in the other direction, see copy_to_host_kernel)
    if (pinned && by_kernel && at.devicePointer && !(((uintptr_t)at.devicePointer | (uintptr_t)d | bytes) & 3)) {
        if (hj_launch_copy_to_host(at.devicePointer, d, bytes, copy) != HJGPU_OK) return fail(ctx, HJGPU_EHIP, "copy_to_host_kernel");
        return HJGPU_OK;
    }
    if (pinned) {
        HIPCHK(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, copy));
        return HJGPU_OK;
    }
    CHK(host_stage(ctx, 2));
    void **stage = ctx->host_stage + 2;
    hipEvent_t *stage_done = ctx->host_stage_ev + 2;
    const size_t stage_bytes = HJ_HOST_STAGE;
    const size_t chunks = (bytes + stage_bytes - 1) / stage_bytes;
    auto len = [&](size_t c) { return c + 1 < chunks ? stage_bytes : bytes - c * stage_bytes; };
    auto fetch = [&](size_t c) -> hipError_t {
        hipError_t e = hipMemcpyAsync(stage[c & 1], (const char *)d + c * stage_bytes, len(c), hipMemcpyDeviceToHost, copy);
        return e != hipSuccess ? e : hipEventRecord(stage_done[c & 1], copy);
    };
    HIPCHK(ctx, fetch(0));
    for (size_t c = 0; c < chunks; ++c) {
        if (c + 1 < chunks) HIPCHK(ctx, fetch(c + 1));            // the other buffer: emptied one round ago
        HIPCHK(ctx, hj_event_synchronize(stage_done[c & 1]));
        memcpy((char *)h + c * stage_bytes, stage[c & 1], len(c));
    }
    return HJGPU_OK;
}

static int join_host_impl(hjgpu_ctx *ctx, int algorithm,
                          const uint32_t *ik, const uint32_t *iv, size_t inner,
                          const uint32_t *ok, const uint32_t *ov, size_t outer,
                          const hjgpu_phj_params *pp, const hjgpu_npj_params *np,
                          const hjgpu_host_rows *rows, hjgpu_result *result, hjgpu_stats *stats, uint64_t *cursor = nullptr);

int hjgpu_join_host(hjgpu_ctx *ctx, int algorithm,
                    const uint32_t *ik, const uint32_t *iv, size_t inner,
                    const uint32_t *ok, const uint32_t *ov, size_t outer,
                    const hjgpu_phj_params *pp, const hjgpu_npj_params *np,
                    hjgpu_result *result, hjgpu_stats *stats)
{
    return join_host_impl(ctx, algorithm, ik, iv, inner, ok, ov, outer, pp, np, nullptr, result, stats);
}

int hjgpu_join_host_rows(hjgpu_ctx *ctx, int algorithm,
                         const uint32_t *ik, const uint32_t *iv, size_t inner,
                         const uint32_t *ok, const uint32_t *ov, size_t outer,
                         const hjgpu_phj_params *pp, const hjgpu_npj_params *np,
                         const hjgpu_host_rows *rows, hjgpu_result *result, hjgpu_stats *stats)
{
    if (!ctx) return HJGPU_EINVAL;
    if (!rows || !result) return fail(ctx, HJGPU_EINVAL, "hjgpu_join_host_rows: rows and result are required");
    if (rows->capacity && (!rows->keys || !rows->outer_vals || !rows->inner_vals))
        return fail(ctx, HJGPU_EINVAL, "hjgpu_join_host_rows: null result column");
    return join_host_impl(ctx, algorithm, ik, iv, inner, ok, ov, outer, pp, np, rows, result, stats);
}

int hjgpu_join_host_rows_shared(hjgpu_ctx *ctx, int algorithm,
                                const uint32_t *ik, const uint32_t *iv, size_t inner,
                                const uint32_t *ok, const uint32_t *ov, size_t outer,
                                const hjgpu_phj_params *pp, const hjgpu_npj_params *np,
                                const hjgpu_host_rows *rows, uint64_t *cursor, hjgpu_result *result, hjgpu_stats *stats)
{
    if (!ctx) return HJGPU_EINVAL;
    if (!rows || !result || !cursor) return fail(ctx, HJGPU_EINVAL, "hjgpu_join_host_rows_shared: rows, cursor and result are required");
    if (rows->capacity && (!rows->keys || !rows->outer_vals || !rows->inner_vals))
        return fail(ctx, HJGPU_EINVAL, "hjgpu_join_host_rows_shared: null result column");
    return join_host_impl(ctx, algorithm, ik, iv, inner, ok, ov, outer, pp, np, rows, result, stats, cursor);
}


// Joins from host columns, aggregates only: the probe side never exists on the device as a whole.  R is uploaded and
// prepared (PHJ / CPRA: hjgpu_phj_build's passes; NPJ: the table, npj.cpp:865-877), then the probe side travels in
// batches of `host_batch` rows through two device buffers: batch i is joined against the prepared build side (K4 .. K8,
// or NPJ's probe, on `run`) while batch i + 1 is on the bus (`copy`).  R join S = union over the batches
// (phj.cpp:1869-1924 runs per partition; the reference's CPRA partitions every chunk of S on its own,
// cpra2.cpp:1757-1827: a batch is such a chunk; an NPJ worker probes its own range of S, npj.cpp:882-901).  What the
// call costs is the upload plus the last batch's join; the device holds R, two batches and a workspace for ONE batch
// (no 8.5 GB columns, no placement search of the twin; a probe side larger than the device's memory is fine).
// Returns HJGPU_OK with *done = false when the call should take the monolithic path instead.
static int join_host_batched(hjgpu_ctx *ctx, int algorithm, const uint32_t *ik, const uint32_t *iv, size_t inner,
                             const uint32_t *ok, const uint32_t *ov, size_t outer, const hjgpu_phj_params *pp,
                             const hjgpu_npj_params *np, const hjgpu_host_rows *rows, hjgpu_result *result, hjgpu_stats *stats,
                             bool *done, uint64_t *cursor = nullptr)
{
    // `cursor` (hjgpu_join_host_rows_shared): the caller's columns are shared by several contexts' calls; a batch's rows go
    // where an atomic fetch-add on *cursor puts them.  Nothing is ever started over then (other calls have appended in
    // between): a batch that does not fit - its device columns or the shared capacity - is counted, not written, and the
    // call returns HJGPU_EOVERFLOW with its exact count.
    *done = false;
    bool shared_overflow = false;
    long long want_batch = ctx->tune.host_batch;
    if (want_batch < 0) {
        // default: rows in batches; aggregates in batches only when whole columns plus their workspace (two packed twins of
        // the probe side: 8 + 16 bytes per probe tuple, 8 + 24 per build tuple) would not fit what is free on the device
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        const double need = 24.0 * (double)outer + 32.0 * (double)inner + 4e9;
        want_batch = (rows || need > (double)free_b) ? (64ll << 20) : 0;
    }
    const size_t B = ((size_t)want_batch + 15) & ~size_t(15);       // rows per batch (batches start 64-byte aligned)
    if (!B || !inner || outer < 2 * B || ctx->tune.batch_tuples) return HJGPU_OK;
    // Materialised rows: every batch's rows are made dense on the device (close_gaps per batch) and travel to
    // the caller's host columns on a third stream while the next batch is joined and the one after it uploaded - PCIe is
    // full duplex, the 12 bytes per result row hide behind the 8 bytes per probe tuple of the upload as far as they can.
    // The per-batch device columns hold the batch's share of rows->capacity with a quarter of headroom: a batch that
    // needs more (or a result beyond the caller's capacity) sends the call down the whole-column path, which knows how
    // to report the needed capacity.
    const size_t nb = (outer + B - 1) / B;
    const u64 row_bs = 4096;
    const size_t workers = !rows ? 0 : algorithm == 0 ? (size_t)hj_npj_probe_grid(ctx->cus, B) * 4
                         : (size_t)std::max(hj_join_workers(ctx->tune, ctx->cus, false, true), hj_join_workers(ctx->tune, ctx->cus, true, true));
    const size_t want_b = rows ? (size_t)((double)rows->capacity * (double)B / (double)outer * 1.25) + row_bs : 0;
    const size_t cap_b = rows ? (want_b / row_bs + 1 + workers) * row_bs : 0;
    void *d_rows[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    hjgpu_output dev_out[2];
    memset(dev_out, 0, sizeof(dev_out));
    DevState *h_state = nullptr;                                     // page-locked: one per batch (dense count, overflow flag)
    hipEvent_t joined[2] = {nullptr, nullptr}, rows_free[2] = {nullptr, nullptr};
    hipStream_t down = nullptr;
    u64 rows_at = 0;                                                 // rows in the caller's columns so far
    bool abandon = false;                                            // take the whole-column path instead
    float ms_download = 0;
    void *d_r[2] = {nullptr, nullptr}, *d_s[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *d_res = nullptr;
    hipEvent_t r_ready = nullptr, s_ready[2] = {nullptr, nullptr}, s_free[2] = {nullptr, nullptr}, b0 = nullptr, b1 = nullptr;
    hipStream_t copy = nullptr, run = nullptr;
    std::vector<hjgpu_result> parts(nb);
    const bool npj = algorithm == 0;
    int rc = HJGPU_OK;
    auto hip_ok = [&](hipError_t e, const char *what) { if (rc == HJGPU_OK && e != hipSuccess) rc = fail(ctx, HJGPU_EHIP, what, e); };
    for (int i = 0; i < 2 && rc == HJGPU_OK; ++i) rc = hjgpu_malloc(ctx, &d_r[i], inner * sizeof(uint32_t));
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 2 && rc == HJGPU_OK; ++i) rc = hjgpu_malloc(ctx, &d_s[s][i], B * sizeof(uint32_t));
    if (rc == HJGPU_OK) rc = hjgpu_malloc(ctx, &d_res, nb * sizeof(hjgpu_result));
    if (rows) {
        for (int s = 0; s < 2; ++s) {
            for (int i = 0; i < 3 && rc == HJGPU_OK; ++i) rc = hjgpu_malloc(ctx, &d_rows[s][i], cap_b * sizeof(uint32_t));
            dev_out[s].d_keys = (uint32_t *)d_rows[s][0]; dev_out[s].d_outer_vals = (uint32_t *)d_rows[s][1];
            dev_out[s].d_inner_vals = (uint32_t *)d_rows[s][2];
            dev_out[s].capacity = cap_b; dev_out[s].block_size = row_bs;
        }
        if (rc == HJGPU_OK && hipHostMalloc(reinterpret_cast<void **>(&h_state), nb * sizeof(DevState), hipHostMallocDefault) != hipSuccess)
            rc = fail(ctx, HJGPU_ENOMEM, "hipHostMalloc(batch states)");
    }
    int least = 0, greatest = 0;
    hip_ok(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    // three priority classes = three hardware-queue pools: upload, join and download never share a queue
    if (!ctx->host_streams[0]) hip_ok(hipStreamCreateWithPriority(&ctx->host_streams[0], hipStreamNonBlocking, greatest), "hipStreamCreate(copy)");
    if (!ctx->host_streams[1]) hip_ok(hipStreamCreateWithFlags(&ctx->host_streams[1], hipStreamNonBlocking), "hipStreamCreate(run)");
    if (!ctx->host_streams[2]) hip_ok(hipStreamCreateWithPriority(&ctx->host_streams[2], hipStreamNonBlocking, least), "hipStreamCreate(down)");
    copy = ctx->host_streams[0]; run = ctx->host_streams[1];
    hip_ok(hipEventCreateWithFlags(&r_ready, hipEventDisableTiming), "hipEventCreate");
    hip_ok(hipEventCreate(&b0), "hipEventCreate"); hip_ok(hipEventCreate(&b1), "hipEventCreate");     // around the build side's work
    // one timed pair per batch: the call's device time is the SUM of the build and of every batch's join (what the
    // reference's programs print is the time of the join, npj.cpp:1104-1114), read once after the pipeline
    std::vector<hipEvent_t> bev(2 * nb, nullptr);
    for (hipEvent_t &e : bev) hip_ok(hipEventCreate(&e), "hipEventCreate");
    for (int b = 0; b < 2; ++b) {
        hip_ok(hipEventCreateWithFlags(&s_ready[b], hipEventDisableTiming), "hipEventCreate");
        hip_ok(hipEventCreateWithFlags(&s_free[b], hipEventDisableTiming), "hipEventCreate");
        if (rows) {
            hip_ok(hipEventCreateWithFlags(&joined[b], hipEventDisableTiming), "hipEventCreate");
            hip_ok(hipEventCreateWithFlags(&rows_free[b], hipEventDisableTiming), "hipEventCreate");
        }
    }
    down = ctx->host_streams[2];
    // A batch whose rows outgrew its device columns (a skewed probe side: most matches in few batches) is joined once more
    // ALONE, into columns made for exactly its rows (its count is exact also when its rows overflowed), and its rows go
    // home from there; returns false when that cannot be done (set below, once the plan exists).
    std::function<bool(size_t, const DevState &, u64)> retry_alone;
    u64 npj_count_before[2] = {0, 0};                                // NPJ: the accumulated count in front of the batch in each slot
    uint32_t retries = 0;
    // batch j's rows -> the caller's columns (its dense count is on the host once `joined` has fired)
    auto download_batch = [&](size_t j) {
        const int slot = (int)(j & 1);
        hip_ok(hj_event_synchronize(joined[slot]), "hipEventSynchronize(joined)");
        if (rc != HJGPU_OK) return;
        const DevState &hs = h_state[j];
        u64 at = rows_at;
        if (cursor) {
            if (hs.overflow) { shared_overflow = true; hip_ok(hipEventRecord(rows_free[slot], down), "hipEventRecord"); return; }
            at = __atomic_fetch_add(cursor, (uint64_t)hs.dense, __ATOMIC_RELAXED);
            if (at + hs.dense > rows->capacity) { shared_overflow = true; hip_ok(hipEventRecord(rows_free[slot], down), "hipEventRecord"); return; }
        } else if (hs.overflow) {
            const u64 need = npj ? hs.result.count - npj_count_before[slot] : hs.result.count;
            if (rows_at + need > rows->capacity || !retry_alone || !retry_alone(j, hs, need)) abandon = true;
            return;
        } else if (rows_at + hs.dense > rows->capacity) { abandon = true; return; }
        const auto d0 = std::chrono::steady_clock::now();
        uint32_t *hcol[3] = {rows->keys, rows->outer_vals, rows->inner_vals};
        for (int i = 0; i < 3 && rc == HJGPU_OK; ++i)
            rc = download_column(ctx, hcol[i] + at, d_rows[slot][i], hs.dense * sizeof(uint32_t), down, true);
        hip_ok(hipEventRecord(rows_free[slot], down), "hipEventRecord");
        rows_at += hs.dense;
        ms_download += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - d0).count();
    };
    PhjPlan pl;
    size_t buckets = 0; uint32_t factor = 0;
    // the workspace (for ONE batch) before the clocks start, like the reference's mamalloc()s (npj.cpp:982-1000 vs 861-863)
    if (rc == HJGPU_OK) {
        const int placement = ctx->tune.placement;          // no placement search in a call that is bound by its upload
        ctx->tune.placement = 1;
        rc = npj ? npj_prepare(ctx, inner, np, &buckets, &factor) : phj_prepare(ctx, inner, B, pp, 1, &pl);
        ctx->tune.placement = placement;
    }
    const bool line = !ctx->tune.npj_refhash, unique = npj && npj_unique(ctx, np);
    DevState *st = reinterpret_cast<DevState *>(ctx->state.p);
    u64 *table = reinterpret_cast<u64 *>(ctx->table.p);
    if (rows && !cursor) retry_alone = [&](size_t j, const DevState &hs, u64 need) -> bool {
        (void)hs;
        const int slot = (int)(j & 1);
        const size_t b = j * B, m = outer - b < B ? outer - b : B;
        const size_t cap = (size_t)((need / row_bs + 1 + workers) * row_bs);
        void *big[3] = {nullptr, nullptr, nullptr};
        hjgpu_result *saved = nullptr;
        bool ok = true;
        for (int i = 0; i < 3 && ok; ++i) ok = hjgpu_malloc(ctx, &big[i], cap * sizeof(uint32_t)) == HJGPU_OK;
        if (ok && npj) ok = hjgpu_malloc(ctx, reinterpret_cast<void **>(&saved), sizeof(hjgpu_result)) == HJGPU_OK;
        hjgpu_output o;
        memset(&o, 0, sizeof(o));
        o.d_keys = (uint32_t *)big[0]; o.d_outer_vals = (uint32_t *)big[1]; o.d_inner_vals = (uint32_t *)big[2];
        o.capacity = cap; o.block_size = row_bs;
        DevState again;
        memset(&again, 0, sizeof(again));
        if (ok && npj) {
            // the accumulated result already holds this batch (counts are exact when rows overflow): what the second run adds is dropped
            ok = hj_copy_async(saved, &st->result, sizeof(hjgpu_result), run) == hipSuccess &&
                 hj_zero_async(&st->block_counter, 3 * sizeof(u64), run) == hipSuccess &&
                 hj_zero_async(&st->overflow, sizeof(uint32_t), run) == hipSuccess &&
                 hj_zero_async(&st->nmoves, sizeof(uint32_t), run) == hipSuccess &&
                 npj_probe_enqueue(ctx, (const uint32_t *)d_s[slot][0], (const uint32_t *)d_s[slot][1], m, table, buckets, factor, &o, run, line, unique) == HJGPU_OK &&
                 hipMemcpyAsync(&again, ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, run) == hipSuccess &&
                 hj_copy_async(&st->result, saved, sizeof(hjgpu_result), run) == hipSuccess;
        } else if (ok) {
            ok = phj_enqueue(ctx, pl, nullptr, nullptr, inner, (const uint32_t *)d_s[slot][0], (const uint32_t *)d_s[slot][1], m, &o, run, nullptr, PHJ_PROBE_ONLY) == HJGPU_OK &&
                 hj_copy_async(static_cast<hjgpu_result *>(d_res) + j, ctx->state.p, sizeof(hjgpu_result), run) == hipSuccess &&
                 hipMemcpyAsync(&again, ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, run) == hipSuccess;
        }
        // the slot's probe rows are needed until here: the next upload into the slot waits for THIS record
        if (hipEventRecord(s_free[slot], run) != hipSuccess) ok = false;
        // Whatever was enqueued above - a copy into the stack variable `again`, kernels that read `saved` and write `big` - has run
        // before this lambda leaves (and frees them), on the failure paths too; NPJ's accumulated result is put back if the chain
        // broke before its restore was enqueued.
        const bool drained = hj_stream_synchronize(run) == hipSuccess;
        if (!ok && npj && saved && drained)
            (void)(hj_copy_async(&st->result, saved, sizeof(hjgpu_result), run) == hipSuccess && hj_stream_synchronize(run) == hipSuccess);
        if (!drained) ok = false;
        if (ok) ok = !again.overflow && again.dense == need;
        if (ok) {
            uint32_t *hcol[3] = {rows->keys, rows->outer_vals, rows->inner_vals};
            for (int i = 0; i < 3 && ok; ++i) ok = download_column(ctx, hcol[i] + rows_at, big[i], need * sizeof(uint32_t), down, true) == HJGPU_OK;
            if (ok) ok = hj_stream_synchronize(down) == hipSuccess;
            if (ok) { rows_at += need; ++retries; }
        }
        (void)hipGetLastError();
        hip_ok(hipEventRecord(rows_free[slot], down), "hipEventRecord");
        for (void *p : big) if (p) (void)hipFree(p);
        if (saved) (void)hipFree(saved);
        return ok;
    };
    float ms_upload = 0;
    if (rc == HJGPU_OK) {
        const auto t0 = std::chrono::steady_clock::now();
        int next = 0;
        rc = upload_column(ctx, d_r[0], ik, inner * sizeof(uint32_t), copy, &next);
        if (rc == HJGPU_OK) rc = upload_column(ctx, d_r[1], iv, inner * sizeof(uint32_t), copy, &next);
        hip_ok(hipEventRecord(r_ready, copy), "hipEventRecord");
        hip_ok(hipStreamWaitEvent(run, r_ready, 0), "hipStreamWaitEvent");
        if (rc == HJGPU_OK && npj) {
            // K1 set() npj.cpp:865-868 ; K2 build() 871-877; the probes of all batches add to ONE result (atomics on the state)
            rc = refuse_capture(ctx, run);
            hip_ok(hipEventRecord(b0, run), "hipEventRecord");
            hip_ok(hj_zero_async(st, sizeof(DevState), run), "clearing the state");
            hip_ok(hj_zero_async(table, buckets * sizeof(u64), run), "clearing the table");
            if (rc == HJGPU_OK) rc = hj_launch_npj_build((const uint32_t *)d_r[0], (const uint32_t *)d_r[1], inner, table, buckets, factor,
                                                        &st->zero_key, ctx->cus, run, line);
            hip_ok(hipEventRecord(b1, run), "hipEventRecord");
        } else if (rc == HJGPU_OK) {
            hip_ok(hipEventRecord(b0, run), "hipEventRecord");
            rc = phj_enqueue(ctx, pl, (const uint32_t *)d_r[0], (const uint32_t *)d_r[1], inner, nullptr, nullptr, 0, nullptr, run, nullptr, PHJ_BUILD_ONLY);
            hip_ok(hipEventRecord(b1, run), "hipEventRecord");
        }
        for (size_t i = 0; i < nb && rc == HJGPU_OK; ++i) {
            const int slot = (int)(i & 1);
            const size_t b = i * B, m = outer - b < B ? outer - b : B;
            if (i >= 2) hip_ok(hipStreamWaitEvent(copy, s_free[slot], 0), "hipStreamWaitEvent");      // batch i - 2 has been joined
            if (rc == HJGPU_OK) rc = upload_column(ctx, d_s[slot][0], ok + b, m * sizeof(uint32_t), copy, &next);
            if (rc == HJGPU_OK) rc = upload_column(ctx, d_s[slot][1], ov + b, m * sizeof(uint32_t), copy, &next);
            hip_ok(hipEventRecord(s_ready[slot], copy), "hipEventRecord");
            hip_ok(hipStreamWaitEvent(run, s_ready[slot], 0), "hipStreamWaitEvent");
            if (rows && i >= 2) hip_ok(hipStreamWaitEvent(run, rows_free[slot], 0), "hipStreamWaitEvent");   // batch i - 2's rows have left
            hip_ok(hipEventRecord(bev[2 * i], run), "hipEventRecord");
            if (rc == HJGPU_OK && npj) {
                // the phase events describe the LAST batch's probe (the build has its own pair)
                for (int e = 0; e < EV_COUNT; ++e) ctx->ev_valid[e] = false;
                record(ctx, EV_BEGIN, run);
                record(ctx, EV_R_HIST, run);
                if (rows) {
                    // the output protocol's counters start over with every batch; the result and the zero-key flag add up
                    hip_ok(hj_zero_async(&st->block_counter, 3 * sizeof(u64), run), "clearing the counters");
                    hip_ok(hj_zero_async(&st->overflow, sizeof(uint32_t), run), "clearing the overflow flag");
                    hip_ok(hj_zero_async(&st->nmoves, sizeof(uint32_t), run), "clearing the move count");
                }
                rc = npj_probe_enqueue(ctx, (const uint32_t *)d_s[slot][0], (const uint32_t *)d_s[slot][1], m, table, buckets, factor,
                                       rows ? &dev_out[slot] : nullptr, run, line, unique);
                ctx->stats.fanout1 = ctx->stats.fanout2 = 0; ctx->stats.buckets = buckets; ctx->last_algo = 0;
                if (rows) {
                    hip_ok(hipMemcpyAsync(&h_state[i], ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, run), "hipMemcpyAsync(state)");
                    hip_ok(hipEventRecord(joined[slot], run), "hipEventRecord");
                }
            } else if (rc == HJGPU_OK) {
                rc = phj_enqueue(ctx, pl, nullptr, nullptr, inner, (const uint32_t *)d_s[slot][0], (const uint32_t *)d_s[slot][1], m,
                                 rows ? &dev_out[slot] : nullptr, run, nullptr, PHJ_PROBE_ONLY);
                hip_ok(hj_copy_async(static_cast<hjgpu_result *>(d_res) + i, ctx->state.p, sizeof(hjgpu_result), run),
                       "hipMemcpyAsync(result)");
                if (rows) {
                    hip_ok(hipMemcpyAsync(&h_state[i], ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, run), "hipMemcpyAsync(state)");
                    hip_ok(hipEventRecord(joined[slot], run), "hipEventRecord");
                }
            }
            hip_ok(hipEventRecord(bev[2 * i + 1], run), "hipEventRecord");
            hip_ok(hipEventRecord(s_free[slot], run), "hipEventRecord");
            // the previous batch's rows go home while this one is joined (its count is on the host by now, or soon)
            if (rows && i >= 1 && rc == HJGPU_OK && !abandon) {
                download_batch(i - 1);
                if (npj) npj_count_before[(int)(i & 1)] = h_state[i - 1].result.count;       // what batch i starts from
            }
            if (abandon) break;
        }
        if (rows && rc == HJGPU_OK && !abandon) download_batch(nb - 1);
        if (rows && rc == HJGPU_OK && !abandon) hip_ok(hj_stream_synchronize(down), "hipStreamSynchronize(down)");
        if (rc == HJGPU_OK) {
            hip_ok(hj_stream_synchronize(copy), "hipStreamSynchronize(copy)");
            ms_upload = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (npj && !abandon) {
                // one accumulated result; key 0 in R -> HJGPU_EZEROKEY (the output counters are the last batch's: not checked here)
                DevState hs;
                hip_ok(hipMemcpyAsync(&hs, ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, run), "hipMemcpyAsync(state)");
                hip_ok(hj_stream_synchronize(run), "hipStreamSynchronize(run)");
                if (rc == HJGPU_OK) {
                    if (result) *result = hs.result;
                    if (hs.zero_key) rc = fail(ctx, HJGPU_EZEROKEY, "NPJ: a build key is 0, the empty-bucket sentinel");
                }
            } else if (!abandon) {
                hip_ok(hipMemcpyAsync(parts.data(), d_res, nb * sizeof(hjgpu_result), hipMemcpyDeviceToHost, run), "hipMemcpyAsync(results)");
                hip_ok(hj_stream_synchronize(run), "hipStreamSynchronize(run)");
            }
        }
    }
    if (!abandon && (rc == HJGPU_OK || (npj && rc == HJGPU_EZEROKEY))) {
        if (!npj) {
            hjgpu_result sum;
            memset(&sum, 0, sizeof(sum));
            for (const hjgpu_result &p : parts) { sum.count += p.count; sum.sum_keys += p.sum_keys; sum.sum_outer_vals += p.sum_outer_vals; sum.sum_inner_vals += p.sum_inner_vals; }
            if (result) *result = sum;
        }
        if (stats) {
            const int rs = hjgpu_get_stats(ctx, stats);          // the phase times of the LAST batch's join ...
            if (rc == HJGPU_OK) rc = rs;
            // ... scaled to the sum over all batches (the batches have one shape; the last may be shorter), plus the
            // build side's work: ms_total is the device time of the whole join, as after a call without batches
            float build_ms = 0, sum = 0, last = 0;
            (void)hipEventElapsedTime(&build_ms, b0, b1);
            for (size_t i = 0; i < nb; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, bev[2 * i], bev[2 * i + 1]) == hipSuccess) { sum += ms; last = ms; } }
            const float k = last > 0 ? sum / last : 1.0f;
            stats->ms_histogram *= k; stats->ms_plan *= k; stats->ms_scatter1 *= k; stats->ms_scatter2 *= k;
            stats->ms_join *= k; stats->ms_close_gaps *= k;
            if (npj) stats->ms_build = build_ms;
            stats->ms_total = build_ms + sum;
            stats->ms_upload = ms_upload; stats->ms_download = ms_download;
            stats->batches = (uint32_t)nb;
        }
        *done = true;
        if (shared_overflow && rc == HJGPU_OK)
            rc = fail(ctx, HJGPU_EOVERFLOW, "hjgpu_join_host_rows_shared: rows of this call did not fit (a batch's device columns or the shared capacity); result->count is exact");
    }
    (void)hipDeviceSynchronize();
    ctx->prepared = false;                             // the build columns are about to be freed with everything else
    for (int s2 = 0; s2 < 2; ++s2) for (int i = 0; i < 3; ++i) if (d_rows[s2][i]) (void)hipFree(d_rows[s2][i]);
    if (h_state) (void)hipHostFree(h_state);
    for (int b = 0; b < 2; ++b) { if (joined[b]) (void)hipEventDestroy(joined[b]); if (rows_free[b]) (void)hipEventDestroy(rows_free[b]); }

    for (int i = 0; i < 2; ++i) if (d_r[i]) (void)hipFree(d_r[i]);
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 2; ++i) if (d_s[s][i]) (void)hipFree(d_s[s][i]);
    if (d_res) (void)hipFree(d_res);
    for (int b = 0; b < 2; ++b) {
        if (s_ready[b]) (void)hipEventDestroy(s_ready[b]);
        if (s_free[b]) (void)hipEventDestroy(s_free[b]);
    }
    if (r_ready) (void)hipEventDestroy(r_ready);
    if (b0) (void)hipEventDestroy(b0);
    if (b1) (void)hipEventDestroy(b1);
    for (hipEvent_t e : bev) if (e) (void)hipEventDestroy(e);
    return rc;
}

static int join_host_impl(hjgpu_ctx *ctx, int algorithm,
                          const uint32_t *ik, const uint32_t *iv, size_t inner,
                          const uint32_t *ok, const uint32_t *ov, size_t outer,
                          const hjgpu_phj_params *pp, const hjgpu_npj_params *np,
                          const hjgpu_host_rows *rows, hjgpu_result *result, hjgpu_stats *stats, uint64_t *cursor)
{
    if (!ctx || algorithm < 0 || algorithm > 2) return HJGPU_EINVAL;
    if ((inner && (!ik || !iv)) || (outer && (!ok || !ov))) return fail(ctx, HJGPU_EINVAL, "null column");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        // the probe side in batches behind the DMA (join_host_batched), all three algorithms, with or without rows; not taken
        // (done = false): probe sides below two batches, host_batch = 0, and - after part of the work - a materialising call
        // whose result outgrows rows->capacity: the call then starts over on the whole-column path below, which knows how
        // to report the needed capacity.  (A batch whose rows outgrow their share of the capacity x 1.25 - a skewed probe
        // side - is joined once more alone, into device columns made for exactly its rows: nothing starts over.)  CPRA in batches: every batch is ONE chunk (the reference
        // partitions every chunk of S on its own, cpra2.cpp:1757-1827: a batch is such a chunk); stats->batches says so.
        bool done = false;
        hjgpu_result batched_result;
        const int brc = join_host_batched(ctx, algorithm, ik, iv, inner, ok, ov, outer, pp, np, rows, result ? result : &batched_result, stats, &done, cursor);
        if (brc != HJGPU_OK || done) return brc;
    }
    // materialised result: device columns of the caller's capacity plus one open block per worker
    // (the reference sizes its output the same way: 1.05 J + 2T blocks, npj.cpp:997-1000)
    hjgpu_output dev_out;
    memset(&dev_out, 0, sizeof(dev_out));
    void *d_rows[3] = {nullptr, nullptr, nullptr};
    const hjgpu_output *out = nullptr;
    hjgpu_result local_result;
    if (rows && !result) result = &local_result;
    void *d[4] = {nullptr, nullptr, nullptr, nullptr};
    const void *h[4] = {ik, iv, ok, ov};
    const size_t n[4] = {inner, inner, outer, outer};
    hipEvent_t r_ready = nullptr, s_ready = nullptr;
    hipStream_t copy = nullptr, run = nullptr;
    int rc = HJGPU_OK;
    auto hip_ok = [&](hipError_t e, const char *what) { if (rc == HJGPU_OK && e != hipSuccess) rc = fail(ctx, HJGPU_EHIP, what, e); };
    for (int i = 0; i < 4 && rc == HJGPU_OK; ++i) rc = hjgpu_malloc(ctx, &d[i], n[i] * sizeof(uint32_t));
    if (rows && inner && outer) {
        const size_t workers = algorithm == 0 ? (size_t)hj_npj_probe_grid(ctx->cus, outer) * 4
                                              : (size_t)std::max(hj_join_workers(ctx->tune, ctx->cus, false, true), hj_join_workers(ctx->tune, ctx->cus, true, true));
        dev_out.block_size = rows->capacity >= (64u << 20) ? 65536 : 1024;
        dev_out.capacity = (rows->capacity / dev_out.block_size + 1 + workers) * dev_out.block_size;
        for (int i = 0; i < 3 && rc == HJGPU_OK; ++i) rc = hjgpu_malloc(ctx, &d_rows[i], dev_out.capacity * sizeof(uint32_t));
        dev_out.d_keys = (uint32_t *)d_rows[0]; dev_out.d_outer_vals = (uint32_t *)d_rows[1];
        dev_out.d_inner_vals = (uint32_t *)d_rows[2];
        out = &dev_out;
    }
    // the upload stream in the high-priority queue pool: its copies never share a hardware queue with the join's kernels
    int least = 0, greatest = 0;
    hip_ok(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    // The context's OWN streams, made once and kept (as in the batched path): the page-locked staging buffers' events
    // (host_stage_ev) outlive a call, and an event that was last recorded on a stream which has since been destroyed made
    // the runtime's next hipEventSynchronize on it fail at random ("operation not permitted when stream is capturing" /
    // "... on an event last recorded in a capturing stream": it looks at the dead stream) - round 4, seen once the
    // multi-GPU host call ran this path on the ranks' contexts again and again.
    if (!ctx->host_streams[0]) hip_ok(hipStreamCreateWithPriority(&ctx->host_streams[0], hipStreamNonBlocking, greatest), "hipStreamCreate(copy)");
    if (!ctx->host_streams[1]) hip_ok(hipStreamCreateWithFlags(&ctx->host_streams[1], hipStreamNonBlocking), "hipStreamCreate(run)");
    copy = ctx->host_streams[0]; run = ctx->host_streams[1];
    hip_ok(hipEventCreateWithFlags(&r_ready, hipEventDisableTiming), "hipEventCreate");
    hip_ok(hipEventCreateWithFlags(&s_ready, hipEventDisableTiming), "hipEventCreate");
    PhjPlan pl;
    size_t buckets = 0; uint32_t factor = 0;
    if (rc == HJGPU_OK) {
        // allocate the workspace before the clocks start, like the reference's mamalloc()s before
        // its timed region (npj.cpp:982-1000 vs 861-863); with the placement search: what the host programs print is the
        // device time of the join, and the twin's placement is 0.4 ms of it
        if (algorithm == 0) rc = npj_prepare(ctx, inner, np, &buckets, &factor);
        else rc = phj_prepare(ctx, inner, outer, pp, algorithm == 2 ? ((pp && pp->chunks) ? pp->chunks : 8) : 1, &pl);
    }
    float ms_upload = 0;
    if (rc == HJGPU_OK) {
        const auto t0 = std::chrono::steady_clock::now();
        int next = 0;
        // probe side first, build side behind it: PHJ / CPRA partition S while R is still arriving
        const int order[4] = {2, 3, 0, 1};
        for (int k = 0; k < 4 && rc == HJGPU_OK; ++k) {
            const int i = order[k];
            rc = upload_column(ctx, d[i], h[i], n[i] * sizeof(uint32_t), copy, &next);
            if (rc == HJGPU_OK && i == 3) hip_ok(hipEventRecord(s_ready, copy), "hipEventRecord");
        }
        hip_ok(hipEventRecord(r_ready, copy), "hipEventRecord");
        const uint32_t *rk = (const uint32_t *)d[0], *rv = (const uint32_t *)d[1];
        const uint32_t *sk = (const uint32_t *)d[2], *sv = (const uint32_t *)d[3];
        if (rc == HJGPU_OK) {
            if (algorithm == 0) {
                // NPJ builds first: it needs R, which arrives last
                hip_ok(hipStreamWaitEvent(run, r_ready, 0), "hipStreamWaitEvent");
                if (rc == HJGPU_OK) rc = npj_enqueue(ctx, rk, rv, inner, sk, sv, outer, buckets, factor, out, run, npj_unique(ctx, np));
            } else {
                hip_ok(hipStreamWaitEvent(run, s_ready, 0), "hipStreamWaitEvent");
                if (rc == HJGPU_OK) rc = phj_enqueue(ctx, pl, rk, rv, inner, sk, sv, outer, out, run, r_ready);
            }
        }
        if (rc == HJGPU_OK) {
            hip_ok(hj_stream_synchronize(copy), "hipStreamSynchronize(copy)");
            ms_upload = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            rc = finish_blocking(ctx, result, out, run);
        }
    }
    // the dense prefix [0, J) of the three result columns -> the caller's host columns
    float ms_download = 0;
    if (rc == HJGPU_OK && rows) {
        // shared columns (hjgpu_join_host_rows_shared): this call's rows go where the cursor puts them
        const u64 at = cursor ? __atomic_fetch_add(cursor, (uint64_t)result->count, __ATOMIC_RELAXED) : 0;
        if (at + result->count > rows->capacity) {
            rc = fail(ctx, HJGPU_EOVERFLOW, "hjgpu_join_host_rows: the result has more rows than rows->capacity (see result->count)");
        } else if (result->count) {
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t *hcol[3] = {rows->keys + at, rows->outer_vals + at, rows->inner_vals + at};
            for (int i = 0; i < 3 && rc == HJGPU_OK; ++i)
                rc = download_column(ctx, hcol[i], d_rows[i], result->count * sizeof(uint32_t), copy);
            if (rc == HJGPU_OK) hip_ok(hj_stream_synchronize(copy), "hipStreamSynchronize(copy)");
            ms_download = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    }
    if (stats && (rc == HJGPU_OK || rc == HJGPU_EOVERFLOW)) {
        const int rs = hjgpu_get_stats(ctx, stats);
        if (rc == HJGPU_OK) rc = rs;
        stats->ms_upload = ms_upload; stats->ms_download = ms_download;
    }
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 4; ++i) if (d[i]) (void)hipFree(d[i]);
    for (int i = 0; i < 3; ++i) if (d_rows[i]) (void)hipFree(d_rows[i]);
    if (r_ready) (void)hipEventDestroy(r_ready);
    if (s_ready) (void)hipEventDestroy(s_ready);
    return rc;
}

}  // extern "C"
