// hjgpu_api.hip — the C-ABI of include/hjgpu.h: context, workspace, launch
// sequencing.  This file is the MI355X counterpart of the reference's thread
// orchestration run()/run_hj() (npj.cpp:769-927, phj.cpp:1646-1949,
// cpra2.cpp:1697-1986): phase order, pass planning, factor choice.  Barriers
// between phases become stream order; there is no host round trip inside a join.
#include "hjgpu_ctx.hpp"
#include "hj_device.hpp"

using namespace hjapi;



namespace hjapi {


int fail(hjgpu_ctx *ctx, int status, const char *what, hipError_t e)
{
    if (ctx) {
        if (e != hipSuccess)
            snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, hipGetErrorString(e));
        else
            snprintf(ctx->err, sizeof(ctx->err), "%s", what);
    }
    return status;
}

static int release(hjgpu_ctx *ctx, void *p)
{
    if (!p) return HJGPU_OK;
    HIPCHK(ctx, hipFree(p));
    return HJGPU_OK;
}



// hipEventSynchronize / hipStreamSynchronize of the host pipelines: a wait that the runtime refuses with
// hipErrorStreamCaptureUnsupported although no stream captures (this library refuses capturing streams: refuse_capture)
// is asked again.  (The refusals seen in round 4 had a cause - staging events recorded on per-call streams that were
// destroyed, see join_host_impl - and are gone with it; the retry stays as a guard.)
hipError_t hj_event_synchronize(hipEvent_t ev)
{
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 200; ++attempt) {
        e = hipEventSynchronize(ev);
        if (e != hipErrorStreamCaptureUnsupported) return e;
        (void)hipGetLastError();
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return e;
}
hipError_t hj_stream_synchronize(hipStream_t st)
{
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 200; ++attempt) {
        e = hipStreamSynchronize(st);
        if (e != hipErrorStreamCaptureUnsupported) return e;
        (void)hipGetLastError();
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return e;
}


int ensure(hjgpu_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return HJGPU_OK;
    if (b.p) { CHK(release(ctx, b.p)); b.p = nullptr; b.cap = 0; }
    // round up so that repeated small growth does not reallocate, and keep a
    // 16-byte tail so aligned vector reads of the last elements stay inside
    size_t want = (bytes + 255) / 256 * 256 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return fail(ctx, HJGPU_ENOMEM, "hipMalloc(workspace)", e); }
    b.cap = want;
    return HJGPU_OK;
}

// Placement-aware allocation of a large buffer that K6 pass 1 scatters into.  On one MI355X the same kernel on the
// same data takes 2.95 or 3.4 ms depending on WHICH allocation the 8 GB pass-1 twin of the probe side lives in: the
// offset inside the allocation does not matter, a fresh hipMalloc of the same size does (physical placement;
// tools/alloc_luck.py, profiles/r02_alloc_luck.txt, r02_shift_sweep.txt), and a plain streaming fill of the buffer shows
// the same split - 1.47 to 1.99 ms per 8.5 GB - and predicts it (tools/ubench_placement.hip, profiles/r02_placement.txt).
// So a large twin is chosen among up to `placement` candidate allocations (all held until the choice is made, so that
// every candidate is different memory) by that fill: the first one that fills at >= 5.5 TB/s is taken, else the
// fastest.  Over fresh processes (tools/placement_dist.sh, profiles/r02_placement_dist.txt) pass 1 then takes
// 2.98-3.04 ms with 12 candidates, 2.96-3.14 with 6, 3.0-3.5 with the first allocation.
// This runs when the workspace grows (hjgpu_reserve / first join), never inside a timed join afterwards.
int ensure_placed(hjgpu_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return HJGPU_OK;
    const int tries = ctx->tune.placement;
    // (twins below 1 GiB - the headline's 512 MB build-side twin, a grouped plan's per-group twins - gain nothing from the
    // search: profiles/r04_ab_placed_min.txt)
    if (tries <= 1 || bytes < ((size_t)1 << 30)) return ensure(ctx, b, bytes);
    if (b.p) { HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = (bytes + 255) / 256 * 256 + 256;
    void *cand[16];
    float ms[16], alloc_ms[16];
    int n = 0, best = -1;
    // the probes run on the context's own non-blocking stream: nothing is issued on the legacy NULL stream (which
    // would synchronise every blocking stream of the process, and is not legal while another stream captures)
    if (!ctx->aux) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(ctx, HJGPU_EHIP, "hipEventCreate"); }
    // candidates are held side by side (so that each is different memory): never more than HALF of what is free when
    // the search starts, so that other allocators of the process (several contexts / loopback ranks on one device,
    // torch's caching allocator) keep room
    size_t free0 = 0, total0 = 0;
    if (hipMemGetInfo(&free0, &total0) != hipSuccess) free0 = 0;
    const size_t budget = free0 / 2;
    const auto search_began = std::chrono::steady_clock::now();
    bool timeboxed = false;
    for (; n < tries; ++n) {
        if (n && (size_t)(n + 1) * want > budget) break;                     // no room for another candidate
        // the search's wall-clock budget (option "placement_ms"): the best block so far is good enough (a block of the slow kind
        // costs pass 1 ~15 %; round 4's driver run looked at twelve 8.5 GB candidates for 3 s and found no fast one)
        if (n && ctx->tune.placement_ms > 0 &&
            std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - search_began).count() >= (float)ctx->tune.placement_ms) {
            timeboxed = true;
            break;
        }
        const auto alloc_began = std::chrono::steady_clock::now();
        if (hipMalloc(&cand[n], want) != hipSuccess) { (void)hipGetLastError(); break; }
        alloc_ms[n] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - alloc_began).count();
        ms[n] = 1e30f;
        for (int rep = 0; rep < 2; ++rep) {                                  // the first touch of fresh memory is slower
            float t = 1e30f;
            if (hipEventRecord(e0, ctx->aux) == hipSuccess && hj_launch_fill_probe(cand[n], want, ctx->aux) == HJGPU_OK &&
                hipEventRecord(e1, ctx->aux) == hipSuccess && hipEventSynchronize(e1) == hipSuccess)
                (void)hipEventElapsedTime(&t, e0, e1);
            if (t < ms[n]) ms[n] = t;
        }
        if (best < 0 || ms[n] < ms[best]) best = n;
        if ((double)want / (ms[n] * 1e-3) >= 5.5e12) { ++n; break; }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (best < 0) return fail(ctx, HJGPU_ENOMEM, "hipMalloc(workspace)");
    if (ctx->tune.placement_log) {
        fprintf(stderr, "hjgpu placement: %zu bytes,", want);
        for (int i = 0; i < n; ++i) fprintf(stderr, " %.3f ms%s (hipMalloc %.1f ms)", ms[i], i == best ? "*" : "", alloc_ms[i]);
        fprintf(stderr, " (%.2f TB/s taken)\n", (double)want / (ms[best] * 1e-3) / 1e12);
    }
    for (int i = 0; i < n; ++i) if (i != best) (void)hipFree(cand[i]);
    // what the search cost beyond one plain allocation: everything from its first hipMalloc to the last candidate freed, minus the kept
    // block's own hipMalloc (hjgpu_stats.placement_search_ms; ms_reserve also holds every other allocation of the workspace and, in a
    // fresh process, the first kernel launch's code-object load)
    ctx->placement_search_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - search_began).count() - alloc_ms[best];
    b.p = cand[best]; b.cap = want;
    ctx->placement_tried = (uint32_t)n; ctx->placement_timeboxed = timeboxed ? 1u : 0u;
    ctx->placement_fill_ms = ms[best]; ctx->placement_bytes = want;
    return HJGPU_OK;
}



uint32_t grouped_groups(const hjgpu_ctx *ctx, size_t inner, size_t outer, const hjgpu_phj_params *prm)
{
    if (prm && (prm->fanout1 || prm->fanout2)) return 0;            // an explicit plan is honoured
    const size_t from = (size_t)ctx->tune.group_from, per = (size_t)ctx->tune.group_inner;
    if (!from || inner < from) return 0;
    if (!ctx->tune.group_always) {
        // Does the extra pass pay?  Measured on one MI355X (profiles/r04_grouped_sweep.txt): beyond two passes' reach
        // (HJGPU_MAX_PARTS 16 K-slot tables at 0.85 of their capacity) every further table fill of a partition probes the
        // partition's probe tuples again, ~3.2 ms per 10^9 probe tuples and fill; pass 0 plus what G smaller joins lose to
        // their shorter launches is ~5 ms per 10^9 tuples of BOTH relations.  (The build itself - 6-8 ms per 10^9 build
        // tuples, small work items bound by latency - costs the same either way.)
        const double reach = (double)HJGPU_MAX_PARTS * hj_join_config_big().cap() * 0.85;
        const double fills = (double)inner / reach;
        if ((fills - 1.0) * 3.2 * (double)outer < 1.1 * 5.0 * ((double)inner + (double)outer)) return 0;
    }
    return (uint32_t)std::min<size_t>((inner + per - 1) / per, 192);
}

}  // namespace hjapi (closed for the entry point below)
int hjgpu_grouped_plan(hjgpu_ctx *ctx, size_t inner, size_t outer, const hjgpu_phj_params *params, uint32_t *groups)
{
    if (!ctx || !groups) return HJGPU_EINVAL;
    const uint32_t g = hjapi::grouped_groups(ctx, inner, outer, params);
    *groups = g > 1 ? g : 0;
    return HJGPU_OK;
}
namespace hjapi {

GroupLayout group_layout(uint32_t G)
{
    GroupLayout l;
    l.G = G;
    l.bins = (64 + G - 1) / G;          // never fewer than 64 bins in K4 / K6 (choose_fanout)
    l.F0 = G * l.bins;                  // <= 255: pass 0 keeps the whole-line carry and its large tiles
    return l;
}

int grouped_twins(hjgpu_ctx *ctx, const GroupLayout &l, size_t inner, size_t outer)
{
    const size_t pad = (size_t)32 * (l.G + 2) + 64;
    CHK(ensure_placed(ctx, ctx->grp[0], (inner + pad) * sizeof(uint32_t)));
    CHK(ensure_placed(ctx, ctx->grp[1], (inner + pad) * sizeof(uint32_t)));
    CHK(ensure_placed(ctx, ctx->grp[2], (outer + pad) * sizeof(uint32_t)));
    CHK(ensure_placed(ctx, ctx->grp[3], (outer + pad) * sizeof(uint32_t)));
    // the two relations' pass-0 offsets, then (device-planned groups) one descriptor of four words per group
    CHK(ensure(ctx, ctx->grp_off, ((size_t)2 * (l.F0 + 1) + (size_t)4 * l.G + 4) * sizeof(u64)));
    return HJGPU_OK;
}




MetaLayout carve(void *base, uint32_t C, uint32_t F1, uint32_t P, size_t ranges, size_t items_extra, size_t tiles2, size_t batches,
                 size_t tdesc_b_cap)
{
    MetaLayout m;
    u64 *p = reinterpret_cast<u64 *>(base);
    size_t at = 0;
    auto take = [&](size_t n) { u64 *r = p ? p + at : nullptr; at += (n + 1) & ~size_t(1); return r; };
    m.counts[0] = take((size_t)C * P);
    m.counts[1] = take((size_t)C * P);
    m.tickets = reinterpret_cast<uint32_t *>(take(HJ_TICKET_WORDS / 2));     // HJ_TICKET_K4 / _K6 / _MULTI_FILL (hj_internal.hpp)
    m.counts_bytes = at * sizeof(u64);
    for (int r = 0; r < 2; ++r) {
        m.off2[r] = take((size_t)C * P + 1);
        m.end2[r] = take((size_t)C * P);
        m.cur2[r] = take((size_t)C * P);
        m.off1[r] = take((size_t)C * F1 + 1);
        m.cur1[r] = take((size_t)C * F1);
        m.tp1[r] = take(C + 1);
        m.seg1[r] = take(C + 1);
        m.tp2[r] = take((size_t)C * F1 + 1);
        m.seg2[r] = take((size_t)F1 + 1);
        m.more[r] = C > 8 ? take(P) : nullptr;
    }
    m.slice_prefix = take((size_t)P + 1);
    m.slices = take(P);
    m.item_part = reinterpret_cast<uint32_t *>(take(((size_t)P + items_extra + 2) / 2 + 1));
    for (int r = 0; r < 2; ++r) {
        m.range_counts[r] = reinterpret_cast<uint32_t *>(take((ranges * F1 + 1) / 2));
        m.range_base[r] = take(ranges * F1);
    }
    m.tdesc_cap = tiles2;
    for (int r = 0; r < 2; ++r) m.tdesc[r] = reinterpret_cast<uint4 *>(take(tiles2 * 4));
    m.boff = take(batches * ((size_t)F1 + 1));
    m.tp2b = take(batches * ((size_t)F1 + 1));
    m.tdescb = reinterpret_cast<uint4 *>(take(batches * tdesc_b_cap * 4));
    m.btickets = reinterpret_cast<uint32_t *>(take(batches + 1));            // 2 x u32 per batch
    m.btickets_bytes = (batches + 1) * sizeof(u64);
    m.total_bytes = at * sizeof(u64);
    return m;
}

// Pass planning.  The reference sizes partitions for a 6400-tuple L2-resident
// table and derives 1-4 passes of equal fan-out (phj.cpp:1791-1808); here the
// partition size comes from the LDS table (HJ_JOIN_CAP tuples at load 0.5) and
// two passes of ~sqrt(P) reach every |R| the 32768-partition cap allows.
void choose_fanout(const HjTuning &tune, size_t inner, const hjgpu_phj_params *prm, uint32_t *F1, uint32_t *F2, bool *big_tables)
{
    uint32_t f1 = prm ? prm->fanout1 : 0, f2 = prm ? prm->fanout2 : 0;
    *big_tables = false;
    if (f1 == 0) {
        double target = tune.join.cap() * 0.85;   // mean fill; Poisson tail stays below CAP
        double parts = ceil((double)inner / target);
        // more partitions than K4's LDS histogram holds: 16 K-slot tables, half as many partitions
        // (|R| = 128 M x |S| = 2.2 G, join phase 5.2 -> 4.3 ms: no table is filled twice)
        if (parts > HJGPU_MAX_PARTS) {
            *big_tables = true;
            target = hj_join_config_big().cap() * 0.85;
            parts = ceil((double)inner / target);
        }
        // never fewer than 64 partitions: with a handful of bins every lane of K4 / K6 adds to the same
        // few LDS words (|R| = 4000 x 1G: histogram 1.97 ms and scatter 4.26 ms at fan-out 2)
        if (parts < 64) parts = 64;
        if (parts > HJGPU_MAX_PARTS) parts = HJGPU_MAX_PARTS;
        // one pass while whole-line mode fits the LDS (fan-out <= 640): 3.7-3.9 ms per 1G tuples at
        // fan-out 288-403 (12 K-tuple tiles), ~5.4 ms at 512-640 (8 K), against 2 x 3.3 ms for two passes
        if (parts <= 640) { f1 = (uint32_t)parts; f2 = 1; }
        else {
            // pass 1: the power of two nearest to sqrt(parts) (H(key, f, 2^k) is a shift: one multiply
            // less per key in K4 and K6) ...
            f1 = 2;
            while ((double)f1 * f1 * 2.0 < parts) f1 <<= 1;          // f1 ~ sqrt(parts) within a factor sqrt(2)
            if (f1 > 256) f1 = 256;
            // ... but as soon as pass 2 still keeps >= 64 partitions, pass 1 takes 192: its cost is flat while the
            // whole-line carry fits beside a 16 K-tuple tile (F <= 209), and every partition less in pass 2 means
            // longer runs per tile, i.e. fewer < 16-tuple tails written into partial lines.  64 M x 1 G, one process:
            // 128 x 144 8.64 ms (2.98 + 3.20), 160 x 116 8.59, 192 x 96 8.55 (3.00 + 3.07), 208 x 89 8.59,
            // 256 x 72 8.71 (3.21 + 3.00: 12 K-tuple tiles in pass 1), 96 x 192 8.74, 64 x 288 8.96
            if (parts >= 192.0 * 64.0) f1 = 192;
            f2 = (uint32_t)ceil(parts / f1);
            while (f2 > HJGPU_MAX_FANOUT) { f1 <<= 1; f2 = (uint32_t)ceil(parts / f1); }
            while ((u64)f1 * f2 > HJGPU_MAX_PARTS) --f2;
        }
    } else if (f2 == 0) f2 = 1;
    *F1 = f1; *F2 = f2;
}

void record(hjgpu_ctx *ctx, int which, hipStream_t s)
{
    if (ctx->ev_cur) {
        // a group of a device-planned grouped join: its own set, and only the six events its merged plan tells apart (a group costs ~20
        // launches; twelve event records on top of them were a third of the enqueue time of a small group)
        if (which == EV_BEGIN || which == EV_S_HIST || which == EV_S_PLAN || which == EV_S_SC1 || which == EV_S_SC2 || which == EV_JOIN)
            (void)hipEventRecord(ctx->ev_cur[which], s);
        return;
    }
    if (which == EV_BEGIN) { ctx->stats_override = false; ctx->grp_ev_groups = 0; }
    ctx->ev_valid[which] = (hipEventRecord(ctx->ev[which], s) == hipSuccess);
}

// option "audit": this call's record - zeroed on `stream`, its last stage = {sequence number, kind, build rows, probe rows} -
// or nullptr when the option is off.  Stage s of the record is rec + 4 * s.
int audit_begin(hjgpu_ctx *ctx, int kind, size_t inner, size_t outer, hipStream_t stream, u64 **rec)
{
    *rec = nullptr;
    if (!ctx->tune.audit) return HJGPU_OK;
    const size_t words = (size_t)HJ_AUDIT_STAGES * 4;
    CHK(ensure(ctx, ctx->audit, (size_t)HJ_AUDIT_RING * words * sizeof(u64)));
    u64 *r = reinterpret_cast<u64 *>(ctx->audit.p) + (size_t)(ctx->audit_seq % HJ_AUDIT_RING) * words;
    HIPCHK(ctx, hj_zero_async(r, words * sizeof(u64), stream));
    CHK(hj_audit_meta(r + 4 * (HJ_AUDIT_STAGES - 1), ctx->audit_seq, (u64)kind, (u64)inner, (u64)outer, stream));
    ctx->audit_seq += 1;
    ctx->audit_checks.clear();
    *rec = r;
    return HJGPU_OK;
}

int audit_partitions(hjgpu_ctx *ctx, int stage, const u64 *tuples, const u64 *beg, const u64 *end, uint32_t parts, const HjAuditHash &h, u64 *rec,
                     hipStream_t stream)
{
    ctx->audit_checks.push_back(hjgpu_ctx::AuditCheck{stage, tuples, beg, end, parts, h});
    return hj_audit_partitions(tuples, beg, end, parts, h, rec + 4 * stage, ctx->cus, stream);
}

// The enqueue paths are not valid inside a HIP stream capture: a replayed graph of one PHJ step faulted on
// gfx950 / ROCm 7.0 (kernels with > 64 KiB of dynamic LDS among the nodes), so a capturing stream is refused
// instead of handing the caller a graph that may corrupt memory.
int refuse_capture(hjgpu_ctx *ctx, hipStream_t stream)
{
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &status) != hipSuccess) { (void)hipGetLastError(); return HJGPU_OK; }
    if (status != hipStreamCaptureStatusNone)
        return fail(ctx, HJGPU_EINVAL, "the stream is capturing a HIP graph: hjgpu joins cannot be captured");
    return HJGPU_OK;
}

int check_columns(hjgpu_ctx *ctx, const uint32_t *k, const uint32_t *v, size_t n)
{
    if (n == 0) return HJGPU_OK;
    if (!k || !v) return fail(ctx, HJGPU_EINVAL, "null column pointer");
    if (((uintptr_t)k & 15) || ((uintptr_t)v & 15))
        return fail(ctx, HJGPU_EALIGN, "key/payload columns must be 16-byte aligned");
    return HJGPU_OK;
}

int setup_output(hjgpu_ctx *ctx, const hjgpu_output *out, uint32_t workers, u64 *block_size,
                 u64 *block_limit)
{
    *block_size = 0; *block_limit = 0;
    if (!out || !out->d_keys) return HJGPU_OK;
    if (!out->d_outer_vals || !out->d_inner_vals) return fail(ctx, HJGPU_EINVAL, "output columns");
    u64 bs = out->block_size ? out->block_size : 65536;
    if (bs < 256 || (bs & (bs - 1))) return fail(ctx, HJGPU_EINVAL, "block_size must be a power of two >= 256");
    u64 bl = out->capacity / bs;
    if (bl == 0) return fail(ctx, HJGPU_EINVAL, "output capacity below one block");
    if (workers > HJ_MAX_WORKERS) return fail(ctx, HJGPU_EINVAL, "too many workers for close_gaps");
    CHK(ensure(ctx, ctx->final_offsets, (size_t)workers * sizeof(u64)));
    CHK(ensure(ctx, ctx->moves, (size_t)2 * HJ_MAX_WORKERS * 24 + (size_t)HJ_MAX_WORKERS * 8));   // move list + stretch starts
    *block_size = bs; *block_limit = bl;
    return HJGPU_OK;
}

// Tiles per range (a range = the unit that owns private pass-1 write cursors).
// Measured at |S| = 1G, F1 = 136: 1..32 tiles per range all land within run-to-run
// noise for K6, while K4/K5b grow from 1.03+0.08 ms to 1.28+0.41 ms at 1 tile per
// range.  Option "range_tiles" overrides (tuning).
uint32_t range_tiles_for(const HjTuning &tune, u64 max_tiles, uint32_t F1)
{
    const int forced = tune.range_tiles;
    // default: ~4096 ranges per relation - fine-grained enough that concurrently running
    // workgroups write neighbouring regions, coarse enough that K4/K5b stay negligible
    u64 k = forced > 0 ? (u64)forced : (max_tiles + 4095) / 4096;
    // whole-line mode (K6 CARRY) completes a line with the NEXT tile of the same range: give a
    // range up to 8 tiles once there are enough tiles to keep >= 512 ranges
    if (forced <= 0) { const u64 want = max_tiles / 512 < 8 ? max_tiles / 512 : 8; if (k < want) k = want; }
    if (k < 1) k = 1;
    while ((max_tiles + k - 1) / k * F1 > HJ_MAX_RANGE_ENTRIES) k *= 2;
    return (uint32_t)k;
}

uint32_t ranges_of(const HjTuning &tune, u64 max_tiles, uint32_t F1)
{
    const uint32_t k = range_tiles_for(tune, max_tiles, F1);
    return (uint32_t)((max_tiles + k - 1) / k);
}

// Largest ranges_of() over every tile count <= max_tiles.  The tiles per range jump at multiples of 512 and of
// 4096 tiles (range_tiles_for), so a SMALLER relation can have MORE ranges than a larger one (733 ranges at
// 12 M tuples, 550 at 18 M): the per-range tables of a prepared build side (hjgpu_phj_build) are sized for any
// batch up to max_outer, not for max_outer itself.  Between two jumps the count grows with the tiles, so the
// maximum sits at max_tiles or right before / at a jump.
uint32_t ranges_capacity(const HjTuning &tune, u64 max_tiles, uint32_t F1)
{
    uint32_t best = ranges_of(tune, max_tiles, F1);
    for (u64 t = 512; t - 1 <= max_tiles && t <= 8192; t += 512) best = std::max(best, ranges_of(tune, t - 1, F1));
    for (u64 t = 4096; t <= max_tiles; t += 4096) best = std::max(best, ranges_of(tune, t, F1));
    return best;
}

Pass1Geom make_geom(const HjTuning &tune, const void *keys, size_t n, uint32_t C, uint32_t F1, bool out_packed, bool capacity)
{
    // chunk ranges = thread_beg/thread_end with alignment 16 (npj.cpp:516-529; cpra2.cpp:1737-1742)
    Pass1Geom g;
    const size_t part = (n / C) & ~size_t(15);
    g.n = n; g.part = part;
    g.chunks = C;
    g.align = align_of(keys);
    g.tile = (uint32_t)hj_scatter_tile(tune, 1, F1, out_packed);
    u64 max_tiles = 1;
    for (uint32_t c = 0; c < C; ++c) {
        // sizing (keys == nullptr): any alignment of the column may follow, one more tile per chunk
        const u64 t = hj_tiles_of(g.beg(c), g.beg(c + 1), g.align, g.tile) + (keys ? 0 : 1);
        if (t > max_tiles) max_tiles = t;
    }
    g.ranges_per_chunk = capacity ? ranges_capacity(tune, max_tiles, F1) : ranges_of(tune, max_tiles, F1);
    if (!capacity && C > 1) {
        // chunks: as many tiles per range as the WHOLE relation would get (never fewer ranges' worth of table than the
        // capacity above: tiles per range only grow with the tile count) - per-chunk sizing gave 8 chunks of 1/8 G tuples
        // twice the ranges of an unchunked relation, and K4 / K5b 0.16 ms more work
        const uint32_t k = range_tiles_for(tune, max_tiles * C, F1);
        const uint32_t fewer = (uint32_t)((max_tiles + k - 1) / k);
        if (fewer < g.ranges_per_chunk) g.ranges_per_chunk = fewer;
    }
    return g;
}

// ---------------------------------------------------------------------------
// PHJ / CPRA: fused histogram -> plan -> scatter x2 -> LDS join
// ---------------------------------------------------------------------------

int phj_prepare(hjgpu_ctx *ctx, size_t inner, size_t outer, const hjgpu_phj_params *prm,
                uint32_t chunks, PhjPlan *pl, bool pre, int big_override, size_t plan_inner)
{
    ctx->prepared = false;               // the workspace is about to be re-planned (hjgpu_phj_build sets it again)
    ReserveClock clock(ctx);
    pl->C = chunks;
    pl->pre = pre ? 1u : 0u; pl->pre_f1 = 1; pl->pre_F1tot = 1; pl->pre_base = 0;
    pl->unique = ctx->tune.unique || (prm && (prm->flags & HJGPU_FLAG_UNIQUE));
    choose_fanout(ctx->tune, plan_inner ? plan_inner : inner, prm, &pl->F1, &pl->F2, &pl->big_tables);
    if (big_override >= 0) pl->big_tables = big_override != 0;
    if (chunks > 8 && !pre) {
        // More than 8 chunks (the reference takes any #threads, cpra2.cpp:2023): always two passes with line-aligned final
        // partitions - every chunk's pass-2 tiles then write ONE shared region per final partition and the join sees a single
        // piece per partition, as after PHJ's passes (it walks at most 8 pieces of a partition in place).
        if (ctx->tune.dense2) return fail(ctx, HJGPU_EINVAL, "more than 8 chunks need the line-aligned final layout (option dense2 is set)");
        if (pl->F2 <= 1) {
            if (prm && prm->fanout1) return fail(ctx, HJGPU_EINVAL, "more than 8 chunks need a two-pass plan (fanout2 >= 2)");
            const uint32_t parts = pl->F1;              // what one pass would have made: 64 ... 640
            pl->F2 = 2; pl->F1 = (parts + 1) / 2;
        }
    }
    if (pl->unique && !hj_join_config_built(hj_join_config_of(ctx->tune, pl->big_tables), true))
        return fail(ctx, HJGPU_EINVAL, "HJGPU_FLAG_UNIQUE: the join_cfg geometry of this context has no _UNIQUE instance "
                                       "(geometries with one: 512,13,2 and 1024,14,2)");
    pl->P = pl->F1 * pl->F2;
    if (pl->F1 < 1 || pl->F2 < 1 || pl->F1 > HJGPU_MAX_FANOUT || pl->F2 > HJGPU_MAX_FANOUT ||
        pl->P < 2 || pl->P > HJGPU_MAX_PARTS)
        return fail(ctx, HJGPU_EINVAL, "fan-out out of range (need 2 <= fanout1*fanout2 <= 32768, each <= 1024)");
    pl->f1 = (prm && prm->factor1) ? prm->factor1 : DEFAULT_F1;
    pl->f2 = (prm && prm->factor2) ? prm->factor2 : DEFAULT_F2;
    pl->tf0 = (prm && prm->table_factor[0]) ? prm->table_factor[0] : DEFAULT_TF0;
    pl->tf1 = (prm && prm->table_factor[1]) ? prm->table_factor[1] : DEFAULT_TF1;
    if (!(pl->f1 & 1) || !(pl->f2 & 1) || !(pl->tf0 & 1) || !(pl->tf1 & 1))
        return fail(ctx, HJGPU_EINVAL, "hash factors must be odd");
    // workspace: pass-1 twins always, pass-2 twins when there is a second pass
    // packed (payload << 32 | key) twins: tmp[0] / tmp[2] = pass-1 output of R / S,
    // tmp[4] / tmp[6] = pass-2 output
    const size_t rb = (inner + 4) * sizeof(u64), sb = (outer + 4) * sizeof(u64);
    if (!pre) {                          // pre-partitioned relations are read where the caller has them: no pass-1 twins
        CHK(ensure(ctx, ctx->tmp[0], rb));
        CHK(ensure_placed(ctx, ctx->tmp[2], sb));
    }
    if (pl->F2 > 1) {
        // the final layout starts every partition on a 128-byte line: < 16 tuples of padding each
        const size_t pad = (size_t)pl->C * pl->P * HJ_LINE_TUPLES * sizeof(u64);
        CHK(ensure(ctx, ctx->tmp[4], rb + pad));
        CHK(ensure(ctx, ctx->tmp[6], sb + pad));
    }
    // ranges of the larger relation bound the per-range tables of both
    // (sized for every relation up to these sizes: a batch of a prepared build side may be smaller than max_outer)
    const Pass1Geom gr = make_geom(ctx->tune, nullptr, inner, pl->C, pl->F1, true, true), gs = make_geom(ctx->tune, nullptr, outer, pl->C, pl->F1, true, true);
    pl->ranges = (size_t)(gr.ranges_per_chunk > gs.ranges_per_chunk ? gr.ranges_per_chunk : gs.ranges_per_chunk) * pl->C;
    pl->items_extra = hj_join_items_capacity(pl->P, outer) - pl->P;
    // pass-2 tiles: whole tiles of the relation plus up to two ragged tiles per segment
    const size_t larger = inner > outer ? inner : outer;
    pl->tiles2 = pl->F2 > 1 ? larger / (size_t)hj_scatter_tile(ctx->tune, 2, pl->F2, true) + 2 * (size_t)pl->C * pl->F1 + 8 : 0;
    // Batched probe-side partitioning (option "batch_tuples" > 0; two-pass, single-chunk plans with line-aligned final
    // partitions): pass 1 of a batch writes into one of two small reused buffers and pass 2 reads it straight away, so
    // the intermediate copy of the probe side can stay in the 256 MiB Infinity Cache.  Plain copies gain from that
    // (tools/ubench_mall_chain.hip: 5.1 ms per 8 GiB through a reused 128-192 MiB buffer against 6.7 ms through an
    // 8 GiB one); K6 does NOT (profiles/r02_batch_sweep.txt: 7.2-9.8 ms for both passes against 6.9 ms unbatched at
    // every batch size, i.e. ~30 us per extra launch and nothing back): K6 is bound by its per-tile work on the CU,
    // not by HBM.  Kept as an option (it also needs no full-size pass-1 twin of the probe side), off by default.
    pl->batch_ranges = pl->batch_cap = pl->batch_tile_cap = 0; pl->tdesc_b_cap = 0; pl->batch_bytes = 0;
    if (pl->F2 > 1 && pl->C == 1 && !ctx->tune.dense2 && ctx->tune.batch_tuples > 0 && pl->F1 <= 1024) {
        const u64 target = (u64)ctx->tune.batch_tuples;
        const u64 tile = gs.tile;
        const u64 tiles = hj_tiles_of(0, outer, 0, (uint32_t)tile) + 1;
        const u64 k = (tiles + ranges_of(ctx->tune, tiles, pl->F1) - 1) / ranges_of(ctx->tune, tiles, pl->F1) + 1;   // tiles per range, with slack
        const u64 rpb = std::max<u64>(1, target / (k * tile));
        const u64 nb = (gs.ranges_per_chunk + rpb - 1) / rpb + 1;
        if (outer >= 3 * target) {
            pl->batch_ranges = (uint32_t)rpb; pl->batch_cap = (uint32_t)nb; pl->batch_tile_cap = (uint32_t)k;
            const u64 bound = rpb * k * tile;                                  // tuples of the largest possible batch
            pl->tdesc_b_cap = (size_t)(bound / (u64)hj_scatter_tile(ctx->tune, 2, pl->F2, true) + 2 * (u64)pl->F1 + 8);
            pl->batch_bytes = (size_t)(bound + 64) * sizeof(u64);
            CHK(ensure(ctx, ctx->tmp[1], pl->batch_bytes));
            CHK(ensure(ctx, ctx->tmp[3], pl->batch_bytes));
        }
    }
    MetaLayout sz = carve(nullptr, pl->C, pl->F1, pl->P, pl->ranges, pl->items_extra, pl->tiles2, pl->batch_cap, pl->tdesc_b_cap);
    CHK(ensure(ctx, ctx->meta, sz.total_bytes));
    CHK(ensure(ctx, ctx->state, sizeof(DevState)));
    return HJGPU_OK;
}

int phj_enqueue(hjgpu_ctx *ctx, const PhjPlan &pl,
                const uint32_t *rk, const uint32_t *rv, size_t inner,
                const uint32_t *sk, const uint32_t *sv, size_t outer,
                const hjgpu_output *out, hipStream_t stream, hipEvent_t inner_ready, PhjMode mode, const PrePieces *pre, const GroupRun *grp)
{
    CHK(refuse_capture(ctx, stream));
    if (grp && (pre || mode != PHJ_WHOLE || inner_ready)) return fail(ctx, HJGPU_EINVAL, "internal: a device-planned group is a whole join on resident columns");
    if ((pre != nullptr) != (pl.pre != 0)) return fail(ctx, HJGPU_EINVAL, "internal: plan and relations disagree about pre-partitioning");
    MetaLayout m = carve(ctx->meta.p, pl.C, pl.F1, pl.P, pl.ranges, pl.items_extra, pl.tiles2, pl.batch_cap, pl.tdesc_b_cap);
    DevState *st = reinterpret_cast<DevState *>(ctx->state.p);
    u64 bs = 0, bl = 0;
    CHK(setup_output(ctx, out, (uint32_t)hj_join_workers(ctx->tune, ctx->cus, pl.big_tables, pl.unique), &bs, &bl));

    record(ctx, EV_BEGIN, stream);
    u64 *audit = nullptr;                // option "audit": this call's record (else NULL: nothing below is enqueued)
    if (!grp) CHK(audit_begin(ctx, (int)mode, inner, outer, stream, &audit));   // (device-planned groups are not audited: option "audit" selects the host-planned form)
    // counts[0] | counts[1] | tickets are contiguous: a whole join zeroes all, a prepared build its own
    // histogram and the tickets, a probe of a prepared build the probe side's histogram and the tickets
    {
        unsigned char *z0 = reinterpret_cast<unsigned char *>(m.counts[0]);
        unsigned char *z1 = reinterpret_cast<unsigned char *>(m.counts[1]);
        unsigned char *zt = reinterpret_cast<unsigned char *>(m.tickets);
        unsigned char *ze = z0 + m.counts_bytes;
        if (mode == PHJ_WHOLE) HIPCHK(ctx, hj_zero_async(z0, m.counts_bytes, stream));
        if (mode == PHJ_BUILD_ONLY) {
            HIPCHK(ctx, hj_zero_async(z0, (size_t)(z1 - z0), stream));
            HIPCHK(ctx, hj_zero_async(zt, (size_t)(ze - zt), stream));
        }
        if (mode == PHJ_PROBE_ONLY) HIPCHK(ctx, hj_zero_async(z1, (size_t)(ze - z1), stream));
    }
    if (mode != PHJ_BUILD_ONLY && !grp) HIPCHK(ctx, hj_zero_async(st, sizeof(DevState), stream));      // (a group: the grouped join's state goes on)

    Pass1Geom geom[2] = {make_geom(ctx->tune, rk, inner, pl.C, pl.F1, true), make_geom(ctx->tune, sk, outer, pl.C, pl.F1, true)};
    if (pre)
        for (int r = 0; r < 2; ++r) {
            if (!pre->tuples[r]) continue;
            geom[r].align = 0;                  // (the pieces' bounds go to the plan kernels as they are: pa.chunk_beg below)
        }
    for (int r = 0; r < 2; ++r)
        if (!pre && (size_t)geom[r].ranges_per_chunk * pl.C > pl.ranges)
            return fail(ctx, HJGPU_EINVAL, "internal: the relation needs more pass-1 ranges than the plan's tables hold");
    const uint32_t *in_k[2] = {rk, sk}, *in_v[2] = {rv, sv};
    const size_t nn[2] = {inner, outer};
    uint32_t *t1[4] = {(uint32_t *)ctx->tmp[0].p, nullptr, (uint32_t *)ctx->tmp[2].p, nullptr};
    uint32_t *t2[4] = {(uint32_t *)ctx->tmp[4].p, nullptr, (uint32_t *)ctx->tmp[6].p, nullptr};
    PlanArgs pa;
    for (int r = 0; r < 2; ++r) {
        pa.counts[r] = m.counts[r]; pa.off2[r] = m.off2[r]; pa.end2[r] = m.end2[r]; pa.cur2[r] = m.cur2[r];
        pa.off1[r] = m.off1[r]; pa.cur1[r] = m.cur1[r]; pa.tp1[r] = m.tp1[r];
        pa.seg1[r] = m.seg1[r]; pa.tp2[r] = m.tp2[r]; pa.tdesc[r] = m.tdesc[r];
        pa.seg2[r] = m.seg2[r];
        pa.more[r] = m.more[r];
    }
    pa.tdesc_cap = (uint32_t)m.tdesc_cap;
    const u64 *dyn[2] = {grp ? grp->desc : nullptr, grp ? grp->desc + 2 : nullptr};           // {first row, rows} of R / S in device memory
    pa.dyn[0] = dyn[0]; pa.dyn[1] = dyn[1];
    pa.unique = pl.unique ? 1u : 0u;
    pa.multi_fill = m.tickets + HJ_TICKET_MULTI_FILL;          // zeroed with the tickets; counted by the work-item plan, read by the _UNIQUE join
    // two-pass plans: final partitions start on 128-byte lines (pass 2 claims whole lines); option "dense2": dense
    const bool pad2 = pl.F2 > 1 && !ctx->tune.dense2;
    pa.pad2 = pad2 ? 1u : 0u;
    pa.n[0] = inner; pa.n[1] = outer;
    // chunked relations (one-GPU CPRA): the chunks' regions of a pass-1 partition side by side, so that pass 2 and the join
    // see what they see after an unchunked pass 1 (the pieces of a pre-partitioned relation lie where they arrived)
    const bool p_major = pl.C > 1 && pad2 && !pre;
    pa.p_major = p_major ? 1u : 0u;
    pa.seg_interleave = (pre && pl.C > 1 && pad2 && ctx->tune.piece_interleave) ? 1u : 0u;
    // pre-partitioned pieces sit at absolute rows [b[0], b[C]) of the caller's array
    if (pre) for (int r = 0; r < 2; ++r) if (pre->tuples[r]) pa.n[r] = pre->ch[r].b[pl.C];
    for (int r = 0; r < 2; ++r) {
        const bool pieces = pre && pre->tuples[r];
        pa.regular[r] = pieces ? 0u : 1u; pa.chunk_part[r] = geom[r].part;
        for (uint32_t c = 0; c < 9; ++c) pa.chunk_beg[r][c] = pieces ? pre->ch[r].b[c <= pl.C ? c : pl.C] : geom[r].beg(c);
    }
    pa.slice_prefix = m.slice_prefix; pa.slices = m.slices; pa.item_part = m.item_part;
    pa.chunks = pl.C; pa.F1 = pl.F1; pa.F2 = pl.F2;
    pa.in_align[0] = pre ? 0u : align_of(rk); pa.in_align[1] = pre ? 0u : align_of(sk);
    pa.tile1 = (uint32_t)hj_scatter_tile(ctx->tune, 1, pl.F1, true); pa.tile2 = (uint32_t)hj_scatter_tile(ctx->tune, 2, pl.F2, true);
    pa.slice = HJ_JOIN_SLICE;
    pa.cap = (uint32_t)hj_join_config_of(ctx->tune, pl.big_tables).cap();

    uint32_t batches_used = 0;
    // option "audit" (audit_kernels.hip): stage 0 / 3 the relation as it is read, 1 / 4 its pass-1 output, 2 / 5 its final partitions
    auto audit_input = [&](int r) -> int {
        if (!audit || !nn[r]) return HJGPU_OK;
        u64 *rec = audit + 4 * (r ? 0 : 3);
        if (pre) return pre->tuples[r] ? hj_audit_sums_packed(pre->tuples[r], pre->ch[r].b[0], pre->ch[r].b[pl.C], rec, ctx->cus, stream) : HJGPU_OK;
        return in_k[r] ? hj_audit_sums_columns(in_k[r], in_v[r], nn[r], rec, ctx->cus, stream) : HJGPU_OK;
    };
    auto audit_pass1 = [&](int r) -> int {
        if (!audit || !nn[r] || pre || pl.C != 1 || pl.F2 <= 1) return HJGPU_OK;
        const HjAuditHash h = {pl.f1, pl.F1, 0u, 1u, 1u, pl.F1};
        return audit_partitions(ctx, r ? 1 : 4, reinterpret_cast<const u64 *>(ctx->tmp[2 * r].p), m.off1[r], nullptr, pl.F1, h, audit, stream);
    };
    auto audit_final = [&](int r) -> int {
        if (!audit || !nn[r]) return HJGPU_OK;
        const bool two = pl.F2 > 1;
        const u64 *fin_r = reinterpret_cast<const u64 *>(two ? ctx->tmp[4 + 2 * r].p : ctx->tmp[2 * r].p);
        const HjAuditHash h = {pre ? pl.pre_f1 : pl.f1, pre ? pl.pre_F1tot : pl.F1, pre ? pl.pre_base : 0u, pl.f2, pl.F2, pl.P};
        return audit_partitions(ctx, r ? 2 : 5, fin_r, m.off2[r], m.end2[r], (pad2 ? 1u : pl.C) * pl.P, h, audit, stream);
    };
    // the stages of one relation's partitioning
    auto k4 = [&](int r) -> int {          // one read of the key column gives the histograms of both passes
        if (nn[r]) CHK(hj_launch_hist2(in_k[r], geom[r], pl.f1, pl.F1, pl.f2, pl.F2, m.counts[r],
                                       m.range_counts[r], m.tickets + HJ_TICKET_K4 + HJ_MAX_CHUNKS * r, ctx->cus, stream, 0, dyn[r]));
        return HJGPU_OK;
    };
    auto k5b = [&](int r) -> int {
        if (nn[r]) CHK(hj_launch_range_base(m.range_counts[r], m.off1[r], m.range_base[r], pl.C,
                                            geom[r].ranges_per_chunk, pl.F1, stream));
        return HJGPU_OK;
    };
    auto pass1 = [&](int r) -> int {       // K6 pass 1: caller's columns -> tmp[0..3]
        if (!nn[r]) return HJGPU_OK;
        ScatterArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.kin = in_k[r]; sa.vin = in_v[r]; sa.kout = t1[2 * r]; sa.vout = t1[2 * r + 1];
        sa.seg_off = m.seg1[r]; sa.tile_prefix = m.tp1[r]; sa.cursors = m.cur1[r];
        sa.nseg = pl.C; sa.F = pl.F1; sa.factor = pl.f1; sa.in_align = align_of(in_k[r]);
        sa.ranged = 1; sa.work_counter = m.tickets + HJ_TICKET_K6 + 2 * r; sa.geom = geom[r]; sa.range_base = m.range_base[r];
        sa.in_packed = 0; sa.out_packed = 1;
        sa.nt_partial = ctx->rows_plain ? 0u : 1u;
        sa.dyn = dyn[r];
        return hj_launch_scatter(sa, ctx->tune, scatter_cus(ctx), stream);
    };
    auto pass2 = [&](int r) -> int {       // K6 pass 2: tmp[0..3] -> tmp[4..7], one segment per (chunk, pass-1 partition)
        if (!nn[r] || pl.F2 <= 1) return HJGPU_OK;
        ScatterArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.kin = t1[2 * r]; sa.vin = t1[2 * r + 1]; sa.kout = t2[2 * r]; sa.vout = t2[2 * r + 1];
        sa.seg_off = p_major ? m.seg2[r] : m.off1[r]; sa.tile_prefix = m.tp2[r]; sa.cursors = m.cur2[r]; sa.tile_desc = m.tdesc[r];
        sa.nseg = p_major ? pl.F1 : pl.C * pl.F1; sa.F = pl.F2; sa.factor = pl.f2; sa.in_align = 0;
        sa.ranged = 0; sa.work_counter = m.tickets + HJ_TICKET_K6 + 2 * r + 1; sa.geom = geom[r]; sa.range_base = nullptr;
        sa.part_start = m.off2[r]; sa.part_end = m.end2[r]; sa.aligned_claims = pad2 ? 1u : 0u;
        sa.in_packed = 1; sa.out_packed = 1;
        sa.nt_partial = ctx->rows_plain ? 0u : 1u;
        return hj_launch_scatter(sa, ctx->tune, scatter_cus(ctx), stream);
    };
    // K4 -> K5 -> K6 x2 for one relation; ev = {after hist, after plan, after pass 1, after pass 2}
    auto partition_relation = [&](int r, uint32_t plan_mask, const int ev[4]) -> int {
        if (pre) {
            // The relation arrives pass-1-partitioned (multi-GPU CPRA: the exchange-level partitioning of the senders'
            // chunks was pass 1, cpra2.cpp:1757-1827): K4p counts per (piece, final partition), K5 lays pass 2 out,
            // K6 pass 2 reads the pieces where they are.  16 + 8 bytes per tuple less than partitioning from scratch.
            if (nn[r] && pre->counts[r])
                HIPCHK(ctx, hj_copy_async(m.counts[r], pre->counts[r], (size_t)pl.C * pl.P * sizeof(u64), stream));
            else if (nn[r]) CHK(hj_launch_hist_packed(pre->tuples[r], pre->ch[r], pl.pre_f1, pl.pre_F1tot, pl.pre_base, pl.F1,
                                                      pl.f2, pl.F2, m.counts[r], ctx->cus, stream));
            record(ctx, ev[0], stream);
            pa.mask = plan_mask;
            CHK(hj_launch_plan(pa, stream));
            record(ctx, ev[1], stream);
            record(ctx, ev[2], stream);
            if (nn[r]) {
                ScatterArgs sa;
                memset(&sa, 0, sizeof(sa));
                sa.kin = reinterpret_cast<const uint32_t *>(pre->tuples[r]); sa.vin = nullptr; sa.kout = t2[2 * r]; sa.vout = t2[2 * r + 1];
                sa.seg_off = m.off1[r]; sa.tile_prefix = m.tp2[r]; sa.cursors = m.cur2[r]; sa.tile_desc = m.tdesc[r];
                sa.nseg = pl.C * pl.F1; sa.F = pl.F2; sa.factor = pl.f2; sa.in_align = 0;
                sa.ranged = 0; sa.work_counter = m.tickets + HJ_TICKET_K6 + 2 * r + 1; sa.geom = geom[r]; sa.range_base = nullptr;
                sa.part_start = m.off2[r]; sa.part_end = m.end2[r]; sa.aligned_claims = pad2 ? 1u : 0u;
                sa.in_packed = 1; sa.out_packed = 1;
                sa.nt_partial = ctx->rows_plain ? 0u : 1u;
                CHK(hj_launch_scatter(sa, ctx->tune, scatter_cus(ctx), stream));
            }
            record(ctx, ev[3], stream);
            CHK(audit_input(r)); CHK(audit_final(r));
            return HJGPU_OK;
        }
        CHK(k4(r));
        record(ctx, ev[0], stream);
        // K5 (+ the join's work items once both histograms exist), K5b
        pa.mask = plan_mask;
        CHK(hj_launch_plan(pa, stream));
        // batched probe side: the tables must hold this relation's batches and tiles per range
        uint32_t batches = 0;
        if (r == 1 && pl.batch_ranges && nn[r]) {
            const u64 tiles = hj_tiles_of(0, nn[r], geom[r].align, geom[r].tile);
            const u64 k = (tiles + geom[r].ranges_per_chunk - 1) / geom[r].ranges_per_chunk;
            const u64 nb = (geom[r].ranges_per_chunk + pl.batch_ranges - 1) / pl.batch_ranges;
            if (k + 1 <= pl.batch_tile_cap && nb <= pl.batch_cap && nb >= 2) batches = (uint32_t)nb;
        }
        if (batches) {
            BatchPlanArgs ba;
            ba.range_counts = m.range_counts[r]; ba.range_base = m.range_base[r]; ba.boff = m.boff; ba.tp2b = m.tp2b;
            ba.tdesc = m.tdescb; ba.tdesc_cap = (uint32_t)pl.tdesc_b_cap; ba.ranges = geom[r].ranges_per_chunk;
            ba.ranges_per_batch = pl.batch_ranges; ba.F1 = pl.F1; ba.F2 = pl.F2;
            ba.tile2 = (uint32_t)hj_scatter_tile(ctx->tune, 2, pl.F2, true);
            CHK(hj_launch_batch_plan(ba, batches, stream));
            HIPCHK(ctx, hj_zero_async(m.btickets, m.btickets_bytes, stream));
            record(ctx, ev[1], stream);
            for (uint32_t b = 0; b < batches; ++b) {
                uint32_t *tbuf = (uint32_t *)ctx->tmp[1 + 2 * (b & 1)].p;
                ScatterArgs sa;
                memset(&sa, 0, sizeof(sa));
                // pass 1 of the batch: the caller's columns -> the batch buffer (dense, from offset 0)
                sa.kin = in_k[r]; sa.vin = in_v[r]; sa.kout = tbuf; sa.vout = nullptr;
                sa.seg_off = m.seg1[r]; sa.tile_prefix = m.tp1[r]; sa.cursors = m.cur1[r];
                sa.nseg = 1; sa.F = pl.F1; sa.factor = pl.f1; sa.in_align = align_of(in_k[r]);
                sa.ranged = 1; sa.work_counter = m.btickets + 2 * b; sa.geom = geom[r]; sa.range_base = m.range_base[r];
                sa.range_begin = b * pl.batch_ranges;
                sa.range_count = std::min(pl.batch_ranges, geom[r].ranges_per_chunk - sa.range_begin);
                sa.in_packed = 0; sa.out_packed = 1;
                sa.nt_partial = ctx->rows_plain ? 0u : 1u;
                CHK(hj_launch_scatter(sa, ctx->tune, scatter_cus(ctx), stream));
                // pass 2 of the batch: the batch buffer -> the relation's final, line-aligned partitions
                memset(&sa, 0, sizeof(sa));
                sa.kin = tbuf; sa.vin = nullptr; sa.kout = t2[2 * r]; sa.vout = t2[2 * r + 1];
                sa.seg_off = m.boff + (size_t)b * (pl.F1 + 1); sa.tile_prefix = m.tp2b + (size_t)b * (pl.F1 + 1);
                sa.cursors = m.cur2[r]; sa.tile_desc = m.tdescb + (size_t)b * pl.tdesc_b_cap * 2;
                sa.nseg = pl.F1; sa.F = pl.F2; sa.factor = pl.f2; sa.in_align = 0;
                sa.ranged = 0; sa.work_counter = m.btickets + 2 * b + 1; sa.geom = geom[r]; sa.range_base = nullptr;
                sa.part_start = m.off2[r]; sa.part_end = m.end2[r]; sa.aligned_claims = 1u;
                sa.in_packed = 1; sa.out_packed = 1;
                sa.nt_partial = ctx->rows_plain ? 0u : 1u;
                CHK(hj_launch_scatter(sa, ctx->tune, scatter_cus(ctx), stream));
            }
            record(ctx, ev[2], stream);         // both passes interleaved: reported as pass 1, pass 2 = 0
            record(ctx, ev[3], stream);
            batches_used = batches;
            CHK(audit_input(r)); CHK(audit_final(r));
            return HJGPU_OK;
        }
        CHK(k5b(r));
        record(ctx, ev[1], stream);
        CHK(pass1(r));
        record(ctx, ev[2], stream);
        CHK(pass2(r));
        record(ctx, ev[3], stream);
        CHK(audit_input(r)); CHK(audit_pass1(r)); CHK(audit_final(r));
        return HJGPU_OK;
    };
    const int ev_s[4] = {EV_S_HIST, EV_S_PLAN, EV_S_SC1, EV_S_SC2};
    const int ev_r[4] = {EV_R_HIST, EV_R_PLAN, EV_R_SC1, EV_R_SC2};
    // Both relations are there from the start (hjgpu_phj / hjgpu_cpra on resident columns: nothing to wait for): their
    // stages run side by side - K4 of R and S, then ONE set of K5 launches that plans both relations and the join's work
    // items (the plan kernels are single-workgroup, latency-bound: two sets cost twice the latency, 0.15 ms per step),
    // then pass 1 of both, then pass 2 of both.  The phase events are recorded at the stage boundaries, so
    // hjgpu_get_stats keeps its meaning (histogram / plan / pass 1 / pass 2 of R and S together).
    const bool merged = grp || (mode == PHJ_WHOLE && !inner_ready && !pre && !pl.batch_ranges && ctx->tune.merged_plan);   // (a group: always)
    if (merged) {
        CHK(k4(0)); CHK(k4(1));
        record(ctx, EV_S_HIST, stream);
        pa.mask = 7u;
        CHK(hj_launch_plan(pa, stream));
        CHK(k5b(0)); CHK(k5b(1));
        record(ctx, EV_S_PLAN, stream);
        CHK(pass1(0)); CHK(pass1(1));
        record(ctx, EV_S_SC1, stream);
        CHK(pass2(0)); CHK(pass2(1));
        record(ctx, EV_S_SC2, stream);
        record(ctx, EV_WAITED, stream);
        for (int e : ev_r) record(ctx, e, stream);
        for (int r = 0; r < 2; ++r) { CHK(audit_input(r)); CHK(audit_pass1(r)); CHK(audit_final(r)); }
    }
    if (!merged && mode != PHJ_BUILD_ONLY) CHK(partition_relation(1, 2u, ev_s));       // probe side first
    else if (!merged) for (int e : ev_s) record(ctx, e, stream);
    if (inner_ready) HIPCHK(ctx, hipStreamWaitEvent(stream, inner_ready, 0));
    if (!merged) record(ctx, EV_WAITED, stream);
    if (!merged && mode == PHJ_WHOLE) CHK(partition_relation(0, 1u | 4u, ev_r));       // build side + join work items
    if (mode == PHJ_BUILD_ONLY) CHK(partition_relation(0, 1u, ev_r));       // build side; work items come with a probe
    if (mode == PHJ_PROBE_ONLY) {
        // the build side was partitioned by hjgpu_phj_build: only the work items are missing
        record(ctx, ev_r[0], stream);
        pa.mask = 4u;
        CHK(hj_launch_plan(pa, stream));
        for (int i = 1; i < 4; ++i) record(ctx, ev_r[i], stream);
        CHK(audit_final(0));               // the prepared build side, as this probe finds it
    }
    const uint32_t *fin[4] = {t1[0], t1[1], t1[2], t1[3]};
    if (pl.F2 > 1) for (int i = 0; i < 4; ++i) fin[i] = t2[i];

    // K7+K8
    if (inner && outer && mode != PHJ_BUILD_ONLY) {
        JoinArgs ja;
        memset(&ja, 0, sizeof(ja));
        ja.rk = fin[0]; ja.rv = fin[1]; ja.sk = fin[2]; ja.sv = fin[3];
        ja.roff = m.off2[0]; ja.soff = m.off2[1];
        ja.rend = m.end2[0]; ja.send = m.end2[1];
        ja.slice_prefix = m.slice_prefix; ja.slices = m.slices; ja.item_part = m.item_part;
        // line-aligned two-pass layout: the chunks' pass-2 tiles wrote every final partition as ONE region
        ja.P = pl.P; ja.chunks = pad2 ? 1u : pl.C;
        ja.f1 = pl.f1; ja.F1 = pl.F1; ja.f2 = pl.f2; ja.F2 = pl.F2;
        if (pre) { ja.f1 = pl.pre_f1; ja.F1 = pl.pre_F1tot; ja.p1_base = pl.pre_base; }   // the empty sentinel of a partition
        ja.tf0 = pl.tf0; ja.tf1 = pl.tf1;
        ja.s_align = 0;
        ja.packed = 1;
        ja.big_tables = pl.big_tables ? 1u : 0u;
        ja.unique = pl.unique ? 1u : 0u;
        ja.result = &st->result;
        // (the work counters live with the tickets: zeroed with them, per join - also for a group, whose DevState is the grouped join's)
        ja.work_counter = reinterpret_cast<u64 *>(m.tickets + HJ_TICKET_JOIN);
        ja.work_counter2 = reinterpret_cast<u64 *>(m.tickets + HJ_TICKET_JOIN2);
        ja.multi_fill = m.tickets + HJ_TICKET_MULTI_FILL;
        ja.resume = grp ? 1u : 0u;
        if (bs) {
            ja.ok = out->d_keys; ja.oov = out->d_outer_vals; ja.oiv = out->d_inner_vals;
            ja.block_size = bs; ja.block_limit = bl;
            ja.block_counter = &st->block_counter;
            ja.final_offsets = (u64 *)ctx->final_offsets.p;
            ja.overflow = &st->overflow;
            ja.nt_rows = ctx->rows_plain ? 0u : 1u;
        }
        CHK(hj_launch_join(ja, ctx->tune, ctx->cus, stream));
        if (audit) CHK(hj_audit_copy(reinterpret_cast<const u64 *>(&st->result), audit + 4 * 6, 4, stream));
    }
    record(ctx, EV_JOIN, stream);
    if (bs && inner && outer && mode != PHJ_BUILD_ONLY && !grp) {
        CHK(hj_launch_close_gaps_ex(out->d_keys, out->d_outer_vals, out->d_inner_vals,
                                    (const u64 *)ctx->final_offsets.p,
                                    (uint32_t)hj_join_workers(ctx->tune, ctx->cus, pl.big_tables, pl.unique), bs, &st->block_counter,
                                    &st->overflow, ctx->moves.p, &st->nmoves, &st->dense, ctx->cus, stream));
    }
    record(ctx, EV_GAPS, stream);
    ctx->stats.fanout1 = pl.F1; ctx->stats.fanout2 = pl.F2; ctx->stats.buckets = 0; ctx->stats.batches = batches_used;
    ctx->last_algo = 1;
    return HJGPU_OK;
}

int finish_blocking(hjgpu_ctx *ctx, hjgpu_result *result, const hjgpu_output *out, hipStream_t stream)
{
    DevState h;
    HIPCHK(ctx, hipMemcpyAsync(&h, ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, stream));
    HIPCHK(ctx, hipStreamSynchronize(stream));
    if (result) *result = h.result;
    if (h.zero_key) return fail(ctx, HJGPU_EZEROKEY, "NPJ: a build key is 0, the empty-bucket sentinel");
    if (h.overflow) return fail(ctx, HJGPU_EOVERFLOW, "materialised output exceeded its capacity");
    if (out && out->d_keys && h.dense != h.result.count && (h.result.count != 0 || h.dense != 0))
        return fail(ctx, HJGPU_EHIP, "internal: dense length != match count after close_gaps");
    return HJGPU_OK;
}

// ---------------------------------------------------------------------------
// NPJ
// ---------------------------------------------------------------------------
bool npj_unique(const hjgpu_ctx *ctx, const hjgpu_npj_params *prm)
{
    return ctx->tune.unique || (prm && (prm->flags & HJGPU_FLAG_UNIQUE));
}

int npj_prepare(hjgpu_ctx *ctx, size_t inner, const hjgpu_npj_params *prm, size_t *buckets,
                uint32_t *factor)
{
    ReserveClock clock(ctx);
    double load = (prm && prm->load > 0) ? prm->load : 0.25;
    if (load > 0.99) return fail(ctx, HJGPU_EINVAL, "load factor must be <= 0.99");
    size_t b = (size_t)((double)inner / load);             // npj.cpp:947
    if (b <= inner) b = inner + 1;
    if (b < 16) b = 16;
    b = (b + 7) & ~size_t(7);                               // whole 64-byte lines of 8 buckets (line-hashed walk)
    *buckets = b;
    *factor = (prm && prm->factor) ? prm->factor : DEFAULT_NPJ_FACTOR;
    if (!(*factor & 1)) return fail(ctx, HJGPU_EINVAL, "hash factor must be odd");
    CHK(ensure_placed(ctx, ctx->table, b * sizeof(u64)));     // >= 1 GB tables: the build (memset + CAS) is 8 % faster in a well-placed block
    CHK(ensure(ctx, ctx->state, sizeof(DevState)));
    return HJGPU_OK;
}

int npj_probe_enqueue(hjgpu_ctx *ctx, const uint32_t *sk, const uint32_t *sv, size_t outer,
                      const u64 *table, size_t buckets, uint32_t factor, const hjgpu_output *out,
                      hipStream_t stream, bool line_hash, bool unique)
{
    DevState *st = reinterpret_cast<DevState *>(ctx->state.p);
    u64 bs = 0, bl = 0;
    const int grid = hj_npj_probe_grid(ctx->cus, outer);
    CHK(setup_output(ctx, out, (uint32_t)grid * 4, &bs, &bl));
    if (outer) {
        NpjProbeArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.keys = sk; pa.vals = sv; pa.n = outer; pa.table = table; pa.buckets = buckets;
        pa.factor = factor; pa.line_hash = line_hash ? 1u : 0u; pa.unique = unique ? 1u : 0u; pa.result = &st->result;
        if (bs) {
            pa.ok = out->d_keys; pa.oov = out->d_outer_vals; pa.oiv = out->d_inner_vals;
            pa.block_size = bs; pa.block_limit = bl; pa.block_counter = &st->block_counter;
            pa.final_offsets = (u64 *)ctx->final_offsets.p; pa.overflow = &st->overflow;
        }
        CHK(hj_launch_npj_probe(pa, ctx->cus, stream, nullptr));
    }
    record(ctx, EV_JOIN, stream);
    if (bs && outer) {
        CHK(hj_launch_close_gaps_ex(out->d_keys, out->d_outer_vals, out->d_inner_vals,
                                    (const u64 *)ctx->final_offsets.p, (uint32_t)grid * 4, bs,
                                    &st->block_counter, &st->overflow, ctx->moves.p, &st->nmoves, &st->dense,
                                    ctx->cus, stream));
    }
    record(ctx, EV_GAPS, stream);
    return HJGPU_OK;
}

int npj_enqueue(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner,
                const uint32_t *sk, const uint32_t *sv, size_t outer,
                size_t buckets, uint32_t factor, const hjgpu_output *out, hipStream_t stream, bool unique)
{
    CHK(refuse_capture(ctx, stream));
    DevState *st = reinterpret_cast<DevState *>(ctx->state.p);
    u64 *table = reinterpret_cast<u64 *>(ctx->table.p);
    record(ctx, EV_BEGIN, stream);
    HIPCHK(ctx, hj_zero_async(st, sizeof(DevState), stream));
    // K1 set() npj.cpp:865-868 ; K2 build() 871-877
    HIPCHK(ctx, hj_zero_async(table, buckets * sizeof(u64), stream));
    // whole joins own their table: line-hashed layout (operator-level hjgpu_npj_build / _probe keep
    // the reference's hash so that their tables stay interchangeable with the reference's)
    const bool line = !ctx->tune.npj_refhash;
    if (inner) CHK(hj_launch_npj_build(rk, rv, inner, table, buckets, factor, &st->zero_key, ctx->cus, stream, line));
    record(ctx, EV_R_HIST, stream);     // reused as "end of build"
    CHK(npj_probe_enqueue(ctx, sk, sv, outer, table, buckets, factor, out, stream, line, unique));
    ctx->stats.fanout1 = ctx->stats.fanout2 = 0; ctx->stats.buckets = buckets; ctx->stats.batches = 0;
    ctx->last_algo = 0;
    return HJGPU_OK;
}

// hjgpu_accumulate_async_status: the two flags of the last join -> two running uint64 counters
__global__ void accumulate_flags_kernel(const DevState *__restrict__ st, u64 *__restrict__ flags)
{
    if (threadIdx.x == 0) { if (st->zero_key) hj_store(&flags[0], flags[0] + 1); if (st->overflow) hj_store(&flags[1], flags[1] + 1); }
}

// the one-shot output of hjgpu_set_async_output
const hjgpu_output *take_async_output(hjgpu_ctx *ctx, const hjgpu_output *given)
{
    if (given) return given;
    if (!ctx->has_pending_out) return nullptr;
    ctx->has_pending_out = false;
    return &ctx->pending_out;
}

}  // namespace hjapi

// ===========================================================================
// extern "C"
// ===========================================================================
static int phj_grouped(hjgpu_ctx *ctx, uint32_t G, uint32_t chunks, const uint32_t *rk, const uint32_t *rv, size_t inner,
                       const uint32_t *sk, const uint32_t *sv, size_t outer, const hjgpu_phj_params *prm, const hjgpu_output *out, hipStream_t stream);
static void grouped_caps(const hjgpu_ctx *ctx, uint32_t G, size_t inner, size_t outer, size_t *cap_r, size_t *cap_s, size_t *plan_inner);

extern "C" {

const char *hjgpu_status_string(int s)
{
    switch (s) {
    case HJGPU_OK: return "ok";
    case HJGPU_EINVAL: return "invalid argument";
    case HJGPU_EALIGN: return "column not 16-byte aligned";
    case HJGPU_ENOMEM: return "out of device memory";
    case HJGPU_EHIP: return "HIP runtime error";
    case HJGPU_EZEROKEY: return "key 0 is reserved by NPJ";
    case HJGPU_EOVERFLOW: return "output capacity exceeded";
    case HJGPU_ENODEVICE: return "no GPU device";
    case HJGPU_ERCCL: return "RCCL / communicator error";
    default: return "unknown status";
    }
}

#ifndef HJGPU_KERNEL_HASH
#define HJGPU_KERNEL_HASH "unknown"
#endif
const char *hjgpu_kernel_hash(void) { return HJGPU_KERNEL_HASH; }
#ifndef HJGPU_LIBRARY_HASH
#define HJGPU_LIBRARY_HASH "unknown"
#endif
const char *hjgpu_library_hash(void) { return HJGPU_LIBRARY_HASH; }

int hjgpu_device_count(int *count)
{
    if (!count) return HJGPU_EINVAL;
    *count = 0;
    if (hipGetDeviceCount(count) != hipSuccess) { *count = 0; (void)hipGetLastError(); return HJGPU_ENODEVICE; }
    return HJGPU_OK;
}

int hjgpu_create(int device, hjgpu_ctx **out)
{
    if (!out) return HJGPU_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return HJGPU_ENODEVICE;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return HJGPU_ENODEVICE; }
    if (device >= n) return HJGPU_EINVAL;
    hjgpu_ctx *ctx = new hjgpu_ctx();
    ctx->err[0] = 0;
    ctx->device = device;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    if (hipSetDevice(device) != hipSuccess ||
        hipGetDeviceProperties(&ctx->prop, device) != hipSuccess) { delete ctx; return HJGPU_ENODEVICE; }
    ctx->cus = ctx->prop.multiProcessorCount;
    hj_tuning_from_env(&ctx->tune);          // the only place the library looks at the environment
    for (int i = 0; i < EV_COUNT; ++i) {
        ctx->ev_valid[i] = false;
        if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { delete ctx; return HJGPU_EHIP; }
    }
    *out = ctx;
    return HJGPU_OK;
}

int hjgpu_destroy(hjgpu_ctx *ctx)
{
    if (!ctx) return HJGPU_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    DevBuf *all[] = {&ctx->tmp[0], &ctx->tmp[1], &ctx->tmp[2], &ctx->tmp[3], &ctx->tmp[4], &ctx->tmp[5],
                     &ctx->tmp[6], &ctx->tmp[7], &ctx->meta, &ctx->table, &ctx->state, &ctx->moves,
                     &ctx->final_offsets, &ctx->grp[0], &ctx->grp[1], &ctx->grp[2], &ctx->grp[3], &ctx->grp_off, &ctx->audit, &ctx->audit_lay};
    for (DevBuf *b : all) if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < EV_COUNT; ++i) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    for (hipEvent_t e : ctx->grp_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->grp_ev_pass0) if (e) (void)hipEventDestroy(e);
    for (int b = 0; b < 4; ++b) {
        if (ctx->host_stage[b]) (void)hipHostFree(ctx->host_stage[b]);
        if (ctx->host_stage_ev[b]) (void)hipEventDestroy(ctx->host_stage_ev[b]);      // (the events go before the streams they were recorded on)
    }
    for (hipStream_t s : ctx->host_streams) if (s) (void)hipStreamDestroy(s);
    delete ctx;
    return HJGPU_OK;
}

const char *hjgpu_last_error(const hjgpu_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int hjgpu_set_option(hjgpu_ctx *ctx, const char *name, const char *value)
{
    if (!ctx || !name || !value) return HJGPU_EINVAL;
    HjTuning t = ctx->tune;
    if (!hj_tuning_set(&t, name, value)) return fail(ctx, HJGPU_EINVAL, "hjgpu_set_option: unknown option or malformed value");
    // the join kernels exist in a fixed set of geometries, _UNIQUE instances in some of them
    if (!hj_join_config_built(t.join, false))
        return fail(ctx, HJGPU_EINVAL, "hjgpu_set_option: join_cfg names a geometry that is not built");
    if (t.unique && !hj_join_config_built(t.join, true))
        return fail(ctx, HJGPU_EINVAL, "hjgpu_set_option: this join_cfg geometry has no _UNIQUE instance (geometries with one: 512,13,2 and 1024,14,2)");
    ctx->tune = t;
    ctx->prepared = false;                   // a prepared build side was planned under the old options
    return HJGPU_OK;
}

int hjgpu_get_device_info(hjgpu_ctx *ctx, hjgpu_device_info *info)
{
    if (!ctx || !info) return HJGPU_EINVAL;
    memset(info, 0, sizeof(*info));
    snprintf(info->name, sizeof(info->name), "%s", ctx->prop.name);
    snprintf(info->arch, sizeof(info->arch), "%s", ctx->prop.gcnArchName);
    info->compute_units = ctx->cus;
    info->lds_bytes_per_block = (int)ctx->prop.sharedMemPerBlock;
    info->hbm_bytes = ctx->prop.totalGlobalMem;
    return HJGPU_OK;
}

int hjgpu_reserve(hjgpu_ctx *ctx, size_t inner, size_t outer)
{
    if (!ctx) return HJGPU_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PhjPlan pl;
    hjgpu_phj_params prm;
    memset(&prm, 0, sizeof(prm));
    const uint32_t groups = grouped_groups(ctx, inner, outer, nullptr);
    if (groups > 1 && outer) {
        // a grouped plan: the pass-0 twins, and the two-pass workspace for ONE group (the largest the device-planned form allows)
        ReserveClock clock(ctx);
        size_t cap_r, cap_s, plan_inner;
        grouped_caps(ctx, groups, inner, outer, &cap_r, &cap_s, &plan_inner);
        CHK(grouped_twins(ctx, group_layout(groups), inner, outer));
        CHK(phj_prepare(ctx, cap_r, cap_s, &prm, 8, &pl, false, -1, plan_inner));
    } else
        CHK(phj_prepare(ctx, inner, outer, &prm, 8, &pl));     // 8 chunks = largest meta
    size_t buckets; uint32_t factor;
    CHK(npj_prepare(ctx, inner, nullptr, &buckets, &factor));
    return HJGPU_OK;
}

// the workspace's side of hjgpu_stats: what the context has spent growing it, and its last placement search
static void fill_reserve(const hjgpu_ctx *ctx, hjgpu_stats *s)
{
    s->ms_reserve = ctx->ms_reserve;
    s->placement_tried = ctx->placement_tried; s->placement_timeboxed = ctx->placement_timeboxed;
    s->placement_fill_ms = ctx->placement_fill_ms; s->placement_search_ms = ctx->placement_search_ms; s->placement_bytes = (uint64_t)ctx->placement_bytes;
}

int hjgpu_get_stats(hjgpu_ctx *ctx, hjgpu_stats *s)
{
    if (!ctx || !s) return HJGPU_EINVAL;
    if (ctx->stats_override) {               // a grouped plan: the sums over pass 0 and the groups' joins, taken as they finished
        *s = ctx->stats;
        fill_reserve(ctx, s);
        return HJGPU_OK;
    }
    if (ctx->grp_ev_groups) {
        // a device-planned grouped join: pass 0, then the groups' phases added up from their own event sets
        HIPCHK(ctx, hipEventSynchronize(ctx->grp_ev_pass0[2]));
        auto between = [&](hipEvent_t a, hipEvent_t b) -> float { float ms = 0; return hipEventElapsedTime(&ms, a, b) == hipSuccess ? ms : 0.f; };
        hjgpu_stats r = ctx->stats;
        r.ms_histogram = r.ms_plan = r.ms_scatter1 = r.ms_scatter2 = r.ms_join = r.ms_close_gaps = r.ms_build = r.ms_inner_wait = 0;
        for (uint32_t g = 0; g < ctx->grp_ev_groups; ++g) {
            const hipEvent_t *e = ctx->grp_ev.data() + (size_t)g * EV_COUNT;
            // (both relations' stages run side by side in a group's plan: phj_enqueue's merged form)
            r.ms_histogram += between(e[EV_BEGIN], e[EV_S_HIST]);
            r.ms_plan += between(e[EV_S_HIST], e[EV_S_PLAN]);
            r.ms_scatter1 += between(e[EV_S_PLAN], e[EV_S_SC1]);
            r.ms_scatter2 += between(e[EV_S_SC1], e[EV_S_SC2]);
            r.ms_join += between(e[EV_S_SC2], e[EV_JOIN]);
        }
        r.ms_scatter0 = between(ctx->grp_ev_pass0[0], ctx->grp_ev_pass0[1]);
        r.ms_total = between(ctx->grp_ev_pass0[0], ctx->grp_ev_pass0[2]);
        r.ms_close_gaps = r.ms_total - (r.ms_scatter0 + r.ms_histogram + r.ms_plan + r.ms_scatter1 + r.ms_scatter2 + r.ms_join);   // close_gaps and the gaps between launches
        if (r.ms_close_gaps < 0) r.ms_close_gaps = 0;
        r.groups = ctx->grp_ev_groups;
        fill_reserve(ctx, &r);
        *s = r;
        return HJGPU_OK;
    }
    if (ctx->ev_valid[EV_GAPS]) HIPCHK(ctx, hipEventSynchronize(ctx->ev[EV_GAPS]));
    auto span = [&](int a, int b) -> float {
        float ms = 0;
        if (ctx->ev_valid[a] && ctx->ev_valid[b] &&
            hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]) == hipSuccess) return ms;
        return 0.f;
    };
    hjgpu_stats r = ctx->stats;
    r.ms_total = span(EV_BEGIN, EV_GAPS);
    r.ms_inner_wait = 0;
    fill_reserve(ctx, &r);
    if (ctx->last_algo == 2) {                 // hjgpu_column_sums: one kernel
        memset(&r, 0, sizeof(r));
        r.ms_total = span(EV_BEGIN, EV_GAPS);
        fill_reserve(ctx, &r);
        *s = r;
        return HJGPU_OK;
    }
    if (ctx->last_algo == 0) {
        r.ms_build = span(EV_BEGIN, EV_R_HIST);
        r.ms_join = span(EV_R_HIST, EV_JOIN);
        r.ms_histogram = r.ms_plan = r.ms_scatter1 = r.ms_scatter2 = 0;
    } else {
        r.ms_histogram = span(EV_BEGIN, EV_S_HIST) + span(EV_WAITED, EV_R_HIST);
        r.ms_plan = span(EV_S_HIST, EV_S_PLAN) + span(EV_R_HIST, EV_R_PLAN);
        r.ms_scatter1 = span(EV_S_PLAN, EV_S_SC1) + span(EV_R_PLAN, EV_R_SC1);
        r.ms_scatter2 = span(EV_S_SC1, EV_S_SC2) + span(EV_R_SC1, EV_R_SC2);
        r.ms_inner_wait = span(EV_S_SC2, EV_WAITED);
        r.ms_join = span(EV_R_SC2, EV_JOIN);
        r.ms_build = 0;
    }
    r.ms_close_gaps = span(EV_JOIN, EV_GAPS);
    r.ms_scatter0 = 0; r.groups = 0;
    *s = r;
    return HJGPU_OK;
}

int hjgpu_get_async_status(hjgpu_ctx *ctx, void *stream_)
{
    if (!ctx) return HJGPU_EINVAL;
    if (!ctx->state.p) return HJGPU_OK;                  // nothing was ever enqueued
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevState h;
    HIPCHK(ctx, hipMemcpyAsync(&h, ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, stream));
    HIPCHK(ctx, hipStreamSynchronize(stream));
    if (ctx->grp_last.valid && h.group_skew) {
        // the device-planned grouped join skipped a group that was larger than its workspace (its d_result says so: all ones): the same
        // join again, host-planned - this thread waits for pass 0 and for every group - with the result where the caller expects it
        const hjgpu_ctx::GroupedCall c = ctx->grp_last;
        ctx->grp_last.valid = false;
        const hjgpu_phj_params *prm = c.has_prm ? &c.prm : nullptr;
        const hjgpu_output *out = c.has_out ? &c.out : nullptr;
        const uint32_t groups = grouped_groups(ctx, c.inner, c.outer, prm);
        if (groups > 1) CHK(phj_grouped(ctx, groups, c.chunks, c.rk, c.rv, c.inner, c.sk, c.sv, c.outer, prm, out, stream));
        else {
            PhjPlan pl;
            CHK(phj_prepare(ctx, c.inner, c.outer, prm, c.chunks, &pl));
            CHK(phj_enqueue(ctx, pl, c.rk, c.rv, c.inner, c.sk, c.sv, c.outer, out, stream));
        }
        if (c.d_result) HIPCHK(ctx, hj_copy_async(c.d_result, ctx->state.p, sizeof(hjgpu_result), stream));
        HIPCHK(ctx, hipMemcpyAsync(&h, ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, stream));
        HIPCHK(ctx, hipStreamSynchronize(stream));
    }
    if (h.zero_key) return fail(ctx, HJGPU_EZEROKEY, "NPJ: a build key is 0, the empty-bucket sentinel");
    if (h.overflow) return fail(ctx, HJGPU_EOVERFLOW, "materialised output exceeded its capacity");
    if (ctx->last_had_output && h.dense != h.result.count && (h.result.count != 0 || h.dense != 0))
        return fail(ctx, HJGPU_EHIP, "internal: dense length != match count after close_gaps");
    return HJGPU_OK;
}

int hjgpu_accumulate_async_status(hjgpu_ctx *ctx, uint64_t *d_flags, void *stream_)
{
    if (!ctx || !d_flags) return HJGPU_EINVAL;
    if (!ctx->state.p) return HJGPU_OK;
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(accumulate_flags_kernel, dim3(1), dim3(64), 0, stream,
                       reinterpret_cast<const DevState *>(ctx->state.p), reinterpret_cast<u64 *>(d_flags));
    HIPCHK(ctx, hipGetLastError());
    return HJGPU_OK;
}

int hjgpu_set_async_output(hjgpu_ctx *ctx, const hjgpu_output *out)
{
    if (!ctx) return HJGPU_EINVAL;
    ctx->has_pending_out = false;
    if (!out || !out->d_keys) return HJGPU_OK;
    if (!out->d_outer_vals || !out->d_inner_vals) return fail(ctx, HJGPU_EINVAL, "output columns");
    ctx->pending_out = *out;
    ctx->has_pending_out = true;
    return HJGPU_OK;
}

int hjgpu_output_capacity(hjgpu_ctx *ctx, int algorithm, size_t outer_tuples, size_t rows, size_t block_size, size_t *capacity)
{
    if (!ctx || !capacity || algorithm < 0 || algorithm > 2) return HJGPU_EINVAL;
    const size_t bs = block_size ? block_size : 65536;
    if (bs < 256 || (bs & (bs - 1))) return fail(ctx, HJGPU_EINVAL, "block_size must be a power of two >= 256");
    const size_t workers = algorithm == 0 ? (size_t)hj_npj_probe_grid(ctx->cus, outer_tuples) * 4
                                          : (size_t)std::max(hj_join_workers(ctx->tune, ctx->cus, false, true), hj_join_workers(ctx->tune, ctx->cus, true, true));
    *capacity = (rows / bs + 1 + workers) * bs;
    return HJGPU_OK;
}

// ---- memory helpers ---------------------------------------------------------
int hjgpu_malloc(hjgpu_ctx *ctx, void **p, size_t bytes)
{
    if (!ctx || !p) return HJGPU_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(p, bytes ? bytes + 16 : 16);   // 16-byte tail for aligned vector reads
    if (e != hipSuccess) { *p = nullptr; return fail(ctx, HJGPU_ENOMEM, "hipMalloc", e); }
    return HJGPU_OK;
}
// A large buffer that a join WRITES into at many places at once - the three result columns of a materialising join -
// is as sensitive to where it lies as the library's own pass-1 twin: the same materialising join of 64 M x 1 G takes
// 3.72 to 4.22 ms depending on the allocation of its result columns (tools/rows_luck.py, profiles/r03_rows_luck.txt).
// The placement search the workspace uses (ensure_placed), for the caller's buffers.
int hjgpu_malloc_placed(hjgpu_ctx *ctx, void **p, size_t bytes)
{
    if (!ctx || !p) return HJGPU_EINVAL;
    *p = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const auto t0 = std::chrono::steady_clock::now();
    DevBuf b;
    CHK(ensure_placed(ctx, b, bytes + 16));
    ctx->ms_reserve += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *p = b.p;
    return HJGPU_OK;
}
int hjgpu_free(hjgpu_ctx *ctx, void *p)
{
    if (!ctx) return HJGPU_EINVAL;
    if (p) HIPCHK(ctx, hipFree(p));
    return HJGPU_OK;
}
int hjgpu_memcpy_h2d(hjgpu_ctx *ctx, void *d, const void *h, size_t bytes)
{
    if (!ctx) return HJGPU_EINVAL;
    if (bytes) HIPCHK(ctx, hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return HJGPU_OK;
}
int hjgpu_memcpy_d2h(hjgpu_ctx *ctx, void *h, const void *d, size_t bytes)
{
    if (!ctx) return HJGPU_EINVAL;
    if (bytes) HIPCHK(ctx, hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    return HJGPU_OK;
}
int hjgpu_host_alloc(hjgpu_ctx *ctx, void **p, size_t bytes)
{
    if (!ctx || !p) return HJGPU_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) { *p = nullptr; return fail(ctx, HJGPU_ENOMEM, "hipHostMalloc", e); }
    return HJGPU_OK;
}
int hjgpu_host_free(hjgpu_ctx *ctx, void *p)
{
    if (!ctx) return HJGPU_EINVAL;
    if (p) HIPCHK(ctx, hipHostFree(p));
    return HJGPU_OK;
}
int hjgpu_audit_read(hjgpu_ctx *ctx, uint64_t *next_seq, uint64_t first_seq, uint32_t count, uint64_t *records, void *stream_)
{
    if (!ctx) return HJGPU_EINVAL;
    if (next_seq) *next_seq = ctx->audit_seq;
    if (!count) return HJGPU_OK;
    if (!records) return fail(ctx, HJGPU_EINVAL, "null records array");
    if (first_seq + count > ctx->audit_seq || ctx->audit_seq - first_seq > (uint64_t)HJ_AUDIT_RING || !ctx->audit.p)
        return fail(ctx, HJGPU_EINVAL, "hjgpu_audit_read: the context keeps the records of its last 256 calls made with option \"audit\"");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_;
    const size_t words = (size_t)HJ_AUDIT_STAGES * 4;
    for (uint32_t i = 0; i < count; ++i)
        HIPCHK(ctx, hipMemcpyAsync(records + (size_t)i * words,
                                   reinterpret_cast<const u64 *>(ctx->audit.p) + (size_t)((first_seq + i) % HJ_AUDIT_RING) * words,
                                   words * sizeof(u64), hipMemcpyDeviceToHost, stream));
    HIPCHK(ctx, hj_stream_synchronize(stream));
    return HJGPU_OK;
}

// Option "audit", second look: the partition checks of the context's LAST audited call once more, with the device quiet -
// (a) by a fresh kernel (through the XCDs' L2s, after a device-wide synchronisation: every L2 written back and invalidated), and
// (b) on the host, from a copy of the buffer made with hipMemcpy (the copy engine reads memory, not an L2).  A stage whose
// first check was wrong and whose memory is wrong here LOST stores; one whose memory is right here was READ STALE.
int hjgpu_audit_recheck(hjgpu_ctx *ctx, uint64_t *words, size_t capacity, size_t *checks)
{
    if (!ctx || !checks) return HJGPU_EINVAL;
    *checks = ctx->audit_checks.size();
    if (!words || capacity < ctx->audit_checks.size()) return HJGPU_OK;      // the caller asks again with room for 9 words per check
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    u64 *d_rec = nullptr;
    HIPCHK(ctx, hipMalloc(&d_rec, 4 * sizeof(u64)));
    if (!ctx->aux) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
    size_t i = 0;
    for (const hjgpu_ctx::AuditCheck &c : ctx->audit_checks) {
        uint64_t *w = words + 9 * i++;
        w[0] = (uint64_t)c.stage;
        u64 fresh[4] = {0, 0, 0, 0};
        HIPCHK(ctx, hj_zero_async(d_rec, sizeof(fresh), ctx->aux));
        CHK(hj_audit_partitions(c.tuples, c.beg, c.end, c.parts, c.h, d_rec, ctx->cus, ctx->aux));
        HIPCHK(ctx, hipMemcpyAsync(fresh, d_rec, sizeof(fresh), hipMemcpyDeviceToHost, ctx->aux));
        HIPCHK(ctx, hipStreamSynchronize(ctx->aux));
        for (int k = 0; k < 4; ++k) w[1 + k] = fresh[k];
        // the host's view: bounds first, then the rows they span
        std::vector<u64> beg(c.parts + 1), end(c.parts);
        HIPCHK(ctx, hipMemcpy(beg.data(), c.beg, (c.end ? c.parts : c.parts + 1) * sizeof(u64), hipMemcpyDeviceToHost));
        if (c.end) HIPCHK(ctx, hipMemcpy(end.data(), c.end, c.parts * sizeof(u64), hipMemcpyDeviceToHost));
        else for (uint32_t q = 0; q < c.parts; ++q) end[q] = beg[q + 1];
        u64 lo = ~0ull, hi = 0;
        for (uint32_t q = 0; q < c.parts; ++q) if (end[q] > beg[q]) { lo = std::min(lo, beg[q]); hi = std::max(hi, end[q]); }
        u64 host[4] = {0, 0, 0, 0};
        if (hi > lo) {
            std::vector<u64> rows(hi - lo);
            HIPCHK(ctx, hipMemcpy(rows.data(), c.tuples + lo, (hi - lo) * sizeof(u64), hipMemcpyDeviceToHost));
            auto H = [](uint32_t key, uint32_t f, uint32_t N) { return (uint32_t)(((u64)(uint32_t)(key * f) * N) >> 32); };
            for (uint32_t q = 0; q < c.parts; ++q) {
                const uint32_t want = q % c.h.modulo;
                for (u64 j = beg[q]; j < end[q]; ++j) {
                    const u64 t = rows[j - lo];
                    const uint32_t key = (uint32_t)t;
                    if ((H(key, c.h.f1, c.h.F1) - c.h.p1_base) * c.h.F2 + H(key, c.h.f2, c.h.F2) != want) ++host[0];
                    host[1] += key; host[2] += t >> 32; ++host[3];
                }
            }
        }
        for (int k = 0; k < 4; ++k) w[5 + k] = host[k];
    }
    HIPCHK(ctx, hipFree(d_rec));
    return HJGPU_OK;
}

int hjgpu_synchronize(hjgpu_ctx *ctx, void *stream)
{
    if (!ctx) return HJGPU_EINVAL;
    HIPCHK(ctx, hipStreamSynchronize((hipStream_t)stream));
    return HJGPU_OK;
}

// ---- whole joins ----------------------------------------------------------------
int hjgpu_npj_async(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner,
                    const uint32_t *sk, const uint32_t *sv, size_t outer,
                    const hjgpu_npj_params *prm, hjgpu_result *d_result, void *stream_)
{
    if (!ctx) return HJGPU_EINVAL;
    // the one-shot output is consumed by THIS call whether it succeeds or not: taken before any early return, so that a
    // failed call never leaves it to an unrelated later join (whose caller may have freed the columns by then)
    const hjgpu_output *out = take_async_output(ctx, nullptr);
    CHK(check_columns(ctx, rk, rv, inner));
    CHK(check_columns(ctx, sk, sv, outer));
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t buckets; uint32_t factor;
    CHK(refuse_capture(ctx, stream));                    // before anything is allocated or probed
    CHK(npj_prepare(ctx, inner, prm, &buckets, &factor));
    ctx->last_had_output = out && out->d_keys;
    CHK(npj_enqueue(ctx, rk, rv, inner, sk, sv, outer, buckets, factor, out, stream, npj_unique(ctx, prm)));
    if (d_result)
        HIPCHK(ctx, hj_copy_async(d_result, ctx->state.p, sizeof(hjgpu_result), stream));
    return HJGPU_OK;
}

int hjgpu_npj(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner,
              const uint32_t *sk, const uint32_t *sv, size_t outer,
              const hjgpu_npj_params *prm, hjgpu_result *result, const hjgpu_output *out, void *stream_)
{
    if (!ctx) return HJGPU_EINVAL;
    CHK(check_columns(ctx, rk, rv, inner));
    CHK(check_columns(ctx, sk, sv, outer));
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t buckets; uint32_t factor;
    CHK(refuse_capture(ctx, stream));
    CHK(npj_prepare(ctx, inner, prm, &buckets, &factor));
    ctx->last_had_output = out && out->d_keys;
    PlainRows plain(ctx, true);
    CHK(npj_enqueue(ctx, rk, rv, inner, sk, sv, outer, buckets, factor, out, stream, npj_unique(ctx, prm)));
    return finish_blocking(ctx, result, out, stream);
}

// Broadcast join: a build side that fits ONE LDS table is not worth partitioning anything.  Every work item (a
// slice of the caller's probe columns) builds the table from the caller's build columns (L2-resident) and
// probes: the probe side is read once (8 B per tuple) instead of K4 + K6 + K7's 28-44 B, and the step is three
// launches.  Same kernel, same cuckoo / chained tables, same block protocol.  |R| = 4000 x |S| = 1 G: 5.4 -> 1.7 ms;
// 1000 x 100 M: 0.76 -> 0.25 ms.  Two fills (8000 rows: every table filled to exactly half, where cuckoo hashing
// gives up and the chained fallback takes over, and the probe side read twice) measured 7.1 ms against 5.5 ms
// for the partitioned plan: one fill only.
// Largest build side of a broadcast join: one fill of the 8 K-slot table, or - up to a load of 0.42, where
// cuckoo insertion still converges quickly - one fill of the 16 K-slot table (one 1024-thread workgroup per CU).
static size_t broadcast_rows(const HjTuning &tune, bool big_tables)
{
    return big_tables ? (size_t)(hj_join_config_big().cap() * 0.85) : (size_t)tune.join.cap();
}

static bool broadcast_applies(const HjTuning &tune, size_t inner, size_t outer, uint32_t chunks, const hjgpu_phj_params *prm)
{
    if (tune.no_broadcast) return false;
    if (chunks != 1 || (prm && (prm->fanout1 || prm->fanout2))) return false;    // an explicit plan is honoured
    return inner && outer && inner <= broadcast_rows(tune, true) && inner <= 16383;
}

static int broadcast_enqueue(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner,
                             const uint32_t *sk, const uint32_t *sv, size_t outer,
                             const hjgpu_phj_params *prm, const hjgpu_output *out, hipStream_t stream,
                             hipEvent_t inner_ready)
{
    CHK(refuse_capture(ctx, stream));
    ctx->prepared = false;
    const uint32_t tf0 = (prm && prm->table_factor[0]) ? prm->table_factor[0] : DEFAULT_TF0;
    const uint32_t tf1 = (prm && prm->table_factor[1]) ? prm->table_factor[1] : DEFAULT_TF1;
    if (!(tf0 & 1) || !(tf1 & 1)) return fail(ctx, HJGPU_EINVAL, "hash factors must be odd");
    const bool big = inner > broadcast_rows(ctx->tune, false);
    const size_t cap = (size_t)hj_join_config_of(ctx->tune, big).cap();
    const size_t nslices = (outer + HJ_JOIN_SLICE - 1) / HJ_JOIN_SLICE;
    const size_t fills = (inner + cap - 1) / cap;
    const bool unique = ctx->tune.unique || (prm && (prm->flags & HJGPU_FLAG_UNIQUE));
    const size_t groups = unique ? 1 : (fills < (size_t)HJ_JOIN_FILL_GROUPS ? fills : (size_t)HJ_JOIN_FILL_GROUPS);
    const size_t items = nslices * groups;
    if (nslices >= (1ull << 32)) return fail(ctx, HJGPU_EINVAL, "probe side too large for a broadcast join");
    // meta: 8 u64 of descriptors, the sentinel, the (all-zero) item directory
    CHK(ensure(ctx, ctx->meta, 16 * sizeof(u64) + (items + 2) * sizeof(uint32_t)));
    CHK(ensure(ctx, ctx->state, sizeof(DevState)));
    u64 *d = reinterpret_cast<u64 *>(ctx->meta.p);
    BroadcastMeta bm;
    bm.roff = d; bm.rend = d + 1; bm.soff = d + 2; bm.send = d + 3; bm.slice_prefix = d + 4; bm.slices = d + 6;
    bm.sentinel = reinterpret_cast<uint32_t *>(d + 8);
    uint32_t *item_part = reinterpret_cast<uint32_t *>(d + 16);
    DevState *st = reinterpret_cast<DevState *>(ctx->state.p);
    u64 bs = 0, bl = 0;
    CHK(setup_output(ctx, out, (uint32_t)hj_join_workers(ctx->tune, ctx->cus, big, unique), &bs, &bl));

    record(ctx, EV_BEGIN, stream);
    HIPCHK(ctx, hj_zero_async(st, sizeof(DevState), stream));
    HIPCHK(ctx, hj_zero_async(item_part, (items + 2) * sizeof(uint32_t), stream));
    for (int e : {EV_S_HIST, EV_S_PLAN, EV_S_SC1, EV_S_SC2}) record(ctx, e, stream);
    if (inner_ready) HIPCHK(ctx, hipStreamWaitEvent(stream, inner_ready, 0));
    record(ctx, EV_WAITED, stream);
    record(ctx, EV_R_HIST, stream);
    CHK(hj_launch_broadcast_meta(rk, inner, outer, (uint32_t)nslices, (uint32_t)groups, bm, stream));
    for (int e : {EV_R_PLAN, EV_R_SC1, EV_R_SC2}) record(ctx, e, stream);
    JoinArgs ja;
    memset(&ja, 0, sizeof(ja));
    ja.rk = rk; ja.rv = rv; ja.sk = sk; ja.sv = sv;
    ja.roff = bm.roff; ja.rend = bm.rend; ja.soff = bm.soff; ja.send = bm.send;
    ja.slice_prefix = bm.slice_prefix; ja.slices = bm.slices; ja.item_part = item_part;
    ja.P = 1; ja.chunks = 1; ja.f1 = ja.f2 = 1; ja.F1 = ja.F2 = 1;
    ja.tf0 = tf0; ja.tf1 = tf1;
    ja.s_align = align_of(sk); ja.packed = 0;
    ja.broadcast = 1; ja.sentinel = bm.sentinel; ja.big_tables = big ? 1u : 0u; ja.unique = unique ? 1u : 0u;
    ja.result = &st->result; ja.work_counter = &st->work_counter; ja.work_counter2 = &st->work_counter2;
    ja.multi_fill = &st->pad;                           // always 0: a broadcast join's build side is one table fill by construction
    if (bs) {
        ja.ok = out->d_keys; ja.oov = out->d_outer_vals; ja.oiv = out->d_inner_vals;
        ja.block_size = bs; ja.block_limit = bl; ja.block_counter = &st->block_counter;
        ja.final_offsets = (u64 *)ctx->final_offsets.p; ja.overflow = &st->overflow;
        ja.nt_rows = ctx->rows_plain ? 0u : 1u;
    }
    CHK(hj_launch_join(ja, ctx->tune, ctx->cus, stream));
    record(ctx, EV_JOIN, stream);
    if (bs)
        CHK(hj_launch_close_gaps_ex(out->d_keys, out->d_outer_vals, out->d_inner_vals,
                                    (const u64 *)ctx->final_offsets.p, (uint32_t)hj_join_workers(ctx->tune, ctx->cus, big, unique), bs,
                                    &st->block_counter, &st->overflow, ctx->moves.p, &st->nmoves, &st->dense,
                                    ctx->cus, stream));
    record(ctx, EV_GAPS, stream);
    ctx->stats.fanout1 = 1; ctx->stats.fanout2 = 1; ctx->stats.buckets = 0; ctx->stats.batches = 0;
    ctx->last_algo = 1;
    return HJGPU_OK;
}

// ---- grouped plans: a third partitioning pass in front of the two-pass join ------------------------------------------
// Two passes end at HJGPU_MAX_PARTS partitions (K4's LDS histogram): with 16 K-slot tables that is a build side of ~228 M
// tuples; beyond it every partition's table is filled several times and the partition's probe tuples are probed once per
// fill (1 G x 1 G: join phase 22 ms of 36, profiles/r04_big_build_before.txt).  The reference adds passes instead
// (1-4 passes of equal fan-out from the partition count, phj.cpp:1791-1808).  Here: PASS 0 splits both relations by an
// independent hash into G key-disjoint groups (the partition operator, fan-out >= 64 in G groups of neighbouring bins, every
// group on a 128-byte line so that its columns can be handed to the two-pass plan as they are), then the groups are joined
// one after the other by the plain plan, and their aggregates (and rows) added up.  One more read + write of both
// relations buys single-fill tables in every group.
static int phj_grouped(hjgpu_ctx *ctx, uint32_t G, uint32_t chunks,
                       const uint32_t *rk, const uint32_t *rv, size_t inner,
                       const uint32_t *sk, const uint32_t *sv, size_t outer,
                       const hjgpu_phj_params *prm, const hjgpu_output *out, hipStream_t stream)
{
    const GroupLayout l = group_layout(G);
    {
        ReserveClock clock(ctx);
        CHK(grouped_twins(ctx, l, inner, outer));
    }
    uint32_t *g_rk = (uint32_t *)ctx->grp[0].p, *g_rv = (uint32_t *)ctx->grp[1].p;
    uint32_t *g_sk = (uint32_t *)ctx->grp[2].p, *g_sv = (uint32_t *)ctx->grp[3].p;
    u64 *d_off = (u64 *)ctx->grp_off.p;
    hjgpu_stats sum, one;
    memset(&sum, 0, sizeof(sum));
    // pass 0: the probe side first, like the join itself (a build side that is still arriving is not supported here)
    // pass 0 must split by a hash that is independent of the groups' own two passes: with the same multiplier every key of a
    // group would fall into 1 / G of the pass-1 partitions (SURVEY appendix A "Factor independence")
    uint32_t f0 = DEFAULT_F0;
    {
        const uint32_t f1 = (prm && prm->factor1) ? prm->factor1 : DEFAULT_F1, f2 = (prm && prm->factor2) ? prm->factor2 : DEFAULT_F2;
        const uint32_t other[3] = {0x7FEB352Du, 0x846CA68Bu, 0xC6A4A793u};
        for (uint32_t cand : other) if (cand != f1 && cand != f2) { f0 = cand; break; }
    }
    CHK(partition_columns(ctx, sk, sv, outer, f0, l.F0, l.bins, g_sk, g_sv, reinterpret_cast<uint64_t *>(d_off + (l.F0 + 1)), stream));
    CHK(hjgpu_get_stats(ctx, &one));
    sum.ms_scatter0 += one.ms_total;
    CHK(partition_columns(ctx, rk, rv, inner, f0, l.F0, l.bins, g_rk, g_rv, reinterpret_cast<uint64_t *>(d_off), stream));
    std::vector<u64> off((size_t)2 * (l.F0 + 1));
    HIPCHK(ctx, hipMemcpyAsync(off.data(), d_off, off.size() * sizeof(u64), hipMemcpyDeviceToHost, stream));
    CHK(hjgpu_get_stats(ctx, &one));                        // waits for the operator's last event
    HIPCHK(ctx, hipStreamSynchronize(stream));
    sum.ms_scatter0 += one.ms_total;
    const u64 *roff = off.data(), *soff = off.data() + (l.F0 + 1);
    struct Piece { u64 r0, rn, s0, sn; };
    std::vector<Piece> pc(G);
    size_t max_r = 0, max_s = 0;
    for (uint32_t g = 0; g < G; ++g) {
        const u64 rb = roff[(size_t)g * l.bins], re = roff[(size_t)(g + 1) * l.bins];
        const u64 sb = soff[(size_t)g * l.bins], se = soff[(size_t)(g + 1) * l.bins];
        pc[g] = Piece{rb + hj_group_shift(rb, g), re - rb, sb + hj_group_shift(sb, g), se - sb};
        max_r = std::max<size_t>(max_r, pc[g].rn); max_s = std::max<size_t>(max_s, pc[g].sn);
    }
    // one workspace for the largest group (no growth, and no placement search, from group to group)
    PhjPlan pl;
    CHK(phj_prepare(ctx, max_r, max_s, prm, chunks, &pl));
    DevState acc;
    memset(&acc, 0, sizeof(acc));
    bool out_on = out && out->d_keys;
    const u64 bs = out_on ? (out->block_size ? out->block_size : 65536) : 0;
    for (uint32_t g = 0; g < G; ++g) {
        if (pc[g].sn == 0 || pc[g].rn == 0) continue;       // nothing can match
        hjgpu_output view;
        const hjgpu_output *vout = nullptr;
        if (out_on) {
            // the group's rows go behind the rows so far: its close_gaps makes [0, count) of the view dense
            const u64 left = out->capacity > acc.dense ? out->capacity - acc.dense : 0;
            if (left / bs == 0) { acc.overflow = 1; out_on = false; }
            else {
                view = *out;
                view.d_keys += acc.dense; view.d_outer_vals += acc.dense; view.d_inner_vals += acc.dense;
                view.capacity = left / bs * bs; view.block_size = bs;
                vout = &view;
            }
        }
        CHK(phj_prepare(ctx, pc[g].rn, pc[g].sn, prm, chunks, &pl));
        CHK(phj_enqueue(ctx, pl, g_rk + pc[g].r0, g_rv + pc[g].r0, pc[g].rn, g_sk + pc[g].s0, g_sv + pc[g].s0, pc[g].sn,
                        vout, stream));
        DevState h;
        HIPCHK(ctx, hipMemcpyAsync(&h, ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, stream));
        HIPCHK(ctx, hipStreamSynchronize(stream));
        CHK(hjgpu_get_stats(ctx, &one));
        sum.ms_histogram += one.ms_histogram; sum.ms_plan += one.ms_plan; sum.ms_scatter1 += one.ms_scatter1;
        sum.ms_scatter2 += one.ms_scatter2; sum.ms_join += one.ms_join; sum.ms_close_gaps += one.ms_close_gaps;
        sum.fanout1 = one.fanout1; sum.fanout2 = one.fanout2;
        acc.result.count += h.result.count; acc.result.sum_keys += h.result.sum_keys;
        acc.result.sum_outer_vals += h.result.sum_outer_vals; acc.result.sum_inner_vals += h.result.sum_inner_vals;
        if (h.overflow) { acc.overflow = 1; out_on = false; }          // the counts stay exact; no rows from here on
        else if (vout) acc.dense += h.dense;
    }
    // the call's result where every entry point looks for it
    // (on the caller's stream, like everything else: nothing is issued on the legacy NULL stream)
    HIPCHK(ctx, hipMemcpyAsync(ctx->state.p, &acc, sizeof(DevState), hipMemcpyHostToDevice, stream));
    HIPCHK(ctx, hipStreamSynchronize(stream));
    sum.ms_total = sum.ms_scatter0 + sum.ms_histogram + sum.ms_plan + sum.ms_scatter1 + sum.ms_scatter2 + sum.ms_join + sum.ms_close_gaps;
    sum.groups = G;
    ctx->stats = sum;
    ctx->stats_override = true;
    ctx->last_algo = 1;
    return HJGPU_OK;
}

// A grouped plan planned ON THE DEVICE (option "group_device", the default): pass 0 of both relations, one small kernel that turns pass 0's
// offsets into a descriptor per group (first row and rows of its build and probe columns inside the pass-0 twins), then G two-pass joins
// enqueued back to back - every kernel of a group's join reads the group's geometry from its descriptor (K4, the plan kernels and K6 pass 1
// take ScatterArgs::dyn / PlanArgs::dyn; everything behind them works on device-resident offsets anyway).  No host thread waits for
// anything: the call is enqueue-only like every other join (phj.cpp:1791-1863 plans and runs its passes inside run_hj).  The groups share
// one join state: aggregates accumulate, the block counter and the waves' open output blocks go on from group to group
// (JoinArgs::resume), ONE close_gaps ends the join.  The workspace is planned for groups of up to (1 + group_slack / 100) x the mean
// group (the fan-out for the mean group: a larger one fills its tables more than once); a group beyond that is skipped and
// DevState::group_skew raised: the caller's next blocking touch point runs the join again in the host-planned form (phj_grouped).
// rows of the largest group a device-planned grouped join has workspace for, and the build rows its fan-out is planned for
static void grouped_caps(const hjgpu_ctx *ctx, uint32_t G, size_t inner, size_t outer, size_t *cap_r, size_t *cap_s, size_t *plan_inner)
{
    const size_t slack = (size_t)ctx->tune.group_slack;
    *cap_r = (inner / G + 1) * (100 + slack) / 100 + 65536;
    *cap_s = (outer / G + 1) * (100 + slack) / 100 + 65536;
    *plan_inner = inner / G + inner / G / 10 + 1;
}

static int phj_grouped_device(hjgpu_ctx *ctx, uint32_t G, uint32_t chunks,
                              const uint32_t *rk, const uint32_t *rv, size_t inner,
                              const uint32_t *sk, const uint32_t *sv, size_t outer,
                              const hjgpu_phj_params *prm, const hjgpu_output *out, hjgpu_result *d_result, hipStream_t stream, hipEvent_t inner_ready)
{
    const GroupLayout l = group_layout(G);
    size_t cap_r, cap_s, plan_inner;
    grouped_caps(ctx, G, inner, outer, &cap_r, &cap_s, &plan_inner);
    PhjPlan pl;
    {
        ReserveClock clock(ctx);
        CHK(grouped_twins(ctx, l, inner, outer));
        CHK(phj_prepare(ctx, cap_r, cap_s, prm, chunks, &pl, false, -1, plan_inner));
        // (a group's plan is the merged one: no batched probe-side partitioning, option "batch_tuples")
        pl.batch_ranges = pl.batch_cap = pl.batch_tile_cap = 0; pl.tdesc_b_cap = 0; pl.batch_bytes = 0;
    }
    // one set of phase events per group (made once, kept): nobody waits between the groups, hjgpu_get_stats adds their spans up afterwards
    while (ctx->grp_ev.size() < (size_t)G * EV_COUNT) {
        hipEvent_t e = nullptr;
        HIPCHK(ctx, hipEventCreate(&e));
        ctx->grp_ev.push_back(e);
    }
    for (hipEvent_t &e : ctx->grp_ev_pass0) if (!e) HIPCHK(ctx, hipEventCreate(&e));
    uint32_t *g_rk = (uint32_t *)ctx->grp[0].p, *g_rv = (uint32_t *)ctx->grp[1].p;
    uint32_t *g_sk = (uint32_t *)ctx->grp[2].p, *g_sv = (uint32_t *)ctx->grp[3].p;
    u64 *d_off = (u64 *)ctx->grp_off.p, *d_desc = d_off + (size_t)2 * (l.F0 + 1);
    DevState *st = reinterpret_cast<DevState *>(ctx->state.p);
    uint32_t f0 = DEFAULT_F0;           // pass 0 splits by a hash independent of the groups' own two passes (see phj_grouped)
    {
        const uint32_t f1 = (prm && prm->factor1) ? prm->factor1 : DEFAULT_F1, f2 = (prm && prm->factor2) ? prm->factor2 : DEFAULT_F2;
        const uint32_t other[3] = {0x7FEB352Du, 0x846CA68Bu, 0xC6A4A793u};
        for (uint32_t cand : other) if (cand != f1 && cand != f2) { f0 = cand; break; }
    }
    const uint32_t workers = (uint32_t)hj_join_workers(ctx->tune, ctx->cus, pl.big_tables, pl.unique);
    u64 bs = 0, bl = 0;
    CHK(setup_output(ctx, out, workers, &bs, &bl));
    HIPCHK(ctx, hipEventRecord(ctx->grp_ev_pass0[0], stream));
    // pass 0: the probe side first; the build side may still be arriving (hjgpu_phj_overlapped_async)
    CHK(partition_columns(ctx, sk, sv, outer, f0, l.F0, l.bins, g_sk, g_sv, reinterpret_cast<uint64_t *>(d_off + (l.F0 + 1)), stream));
    if (inner_ready) HIPCHK(ctx, hipStreamWaitEvent(stream, inner_ready, 0));
    CHK(partition_columns(ctx, rk, rv, inner, f0, l.F0, l.bins, g_rk, g_rv, reinterpret_cast<uint64_t *>(d_off), stream));
    // the grouped join's state: cleared ONCE; every wave's output cursor "no block yet"
    HIPCHK(ctx, hj_zero_async(st, sizeof(DevState), stream));
    if (bs) HIPCHK(ctx, hj_fill_async(ctx->final_offsets.p, 0xFFFFFFFFu, (size_t)workers * sizeof(u64), stream));
    CHK(hj_launch_group_desc(d_off, d_off + (l.F0 + 1), G, l.bins, (u64)cap_r, (u64)cap_s, (u64)inner, (u64)outer, d_desc, &st->group_skew, stream));
    HIPCHK(ctx, hipEventRecord(ctx->grp_ev_pass0[1], stream));
    for (uint32_t g = 0; g < G; ++g) {
        const GroupRun run = {d_desc + 4 * (size_t)g};
        ctx->ev_cur = ctx->grp_ev.data() + (size_t)g * EV_COUNT;
        const int rc = phj_enqueue(ctx, pl, g_rk, g_rv, cap_r, g_sk, g_sv, cap_s, bs ? out : nullptr, stream, nullptr, PHJ_WHOLE, nullptr, &run);
        ctx->ev_cur = nullptr;
        CHK(rc);
    }
    if (bs)
        CHK(hj_launch_close_gaps_ex(out->d_keys, out->d_outer_vals, out->d_inner_vals, (const u64 *)ctx->final_offsets.p, workers, bs,
                                    &st->block_counter, &st->overflow, ctx->moves.p, &st->nmoves, &st->dense, ctx->cus, stream));
    if (d_result) CHK(hj_launch_group_result(&st->result, &st->group_skew, d_result, stream));
    HIPCHK(ctx, hipEventRecord(ctx->grp_ev_pass0[2], stream));
    // the last event every waiter looks at (hjgpu_get_stats), recorded in the context's own set
    for (int i = 0; i < EV_COUNT; ++i) ctx->ev_valid[i] = false;
    record(ctx, EV_BEGIN, stream);
    record(ctx, EV_GAPS, stream);
    ctx->grp_ev_groups = G;
    ctx->stats.fanout1 = pl.F1; ctx->stats.fanout2 = pl.F2; ctx->stats.buckets = 0; ctx->stats.batches = 0;
    ctx->last_algo = 1;
    return HJGPU_OK;
}

// does a grouped plan of this context run device-planned?  (option "audit" reads every stage's output with host-known sizes: host-planned)
static bool grouped_on_device(const hjgpu_ctx *ctx) { return ctx->tune.group_device && !ctx->tune.audit && !ctx->tune.scatter_prof; }

static int phj_like(hjgpu_ctx *ctx, uint32_t chunks,
                    const uint32_t *rk, const uint32_t *rv, size_t inner,
                    const uint32_t *sk, const uint32_t *sv, size_t outer,
                    const hjgpu_phj_params *prm, hjgpu_result *result, hjgpu_result *d_result,
                    const hjgpu_output *out, void *stream_, bool blocking, void *inner_ready = nullptr, bool local_join = false)
{
    if (!ctx) return HJGPU_EINVAL;
    if (!blocking) out = take_async_output(ctx, out);    // consumed by this call even if it fails below (see hjgpu_npj_async)
    PlainRows plain(ctx, blocking);
    CHK(check_columns(ctx, rk, rv, inner));
    CHK(check_columns(ctx, sk, sv, outer));
    if (chunks < 1 || chunks > HJ_MAX_CHUNKS) return fail(ctx, HJGPU_EINVAL, "chunks must be in [1, 256]");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(refuse_capture(ctx, stream));                    // before anything is allocated or probed
    ctx->last_had_output = out && out->d_keys;
    ctx->grp_last.valid = false;
    const uint32_t groups = grouped_groups(ctx, inner, outer, prm);
    // (the local join of a multi-GPU call - hjgpu_phj_overlapped_async - is always planned on the device: a rank's host thread waits with a
    // deadline or not at all)
    if (groups > 1 && outer && (grouped_on_device(ctx) || local_join)) {
        CHK(phj_grouped_device(ctx, groups, chunks, rk, rv, inner, sk, sv, outer, prm, out, d_result, stream, (hipEvent_t)inner_ready));
        if (!blocking) {
            // what hjgpu_get_async_status needs to do the join again, host-planned, should a group have been larger than its workspace
            hjgpu_ctx::GroupedCall &c = ctx->grp_last;
            c.valid = true; c.chunks = chunks; c.rk = rk; c.rv = rv; c.sk = sk; c.sv = sv; c.inner = inner; c.outer = outer;
            c.has_prm = prm != nullptr; c.has_out = out != nullptr; c.d_result = d_result;
            if (prm) c.prm = *prm;
            if (out) c.out = *out;
            return HJGPU_OK;
        }
        DevState h;
        HIPCHK(ctx, hipMemcpyAsync(&h, ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, stream));
        HIPCHK(ctx, hipStreamSynchronize(stream));
        if (!h.group_skew) return finish_blocking(ctx, result, out, stream);
        // a group was larger than the plan's workspace (heavy duplicates): the host-planned form sizes every group's join from its rows
        CHK(phj_grouped(ctx, groups, chunks, rk, rv, inner, sk, sv, outer, prm, out, stream));
        return finish_blocking(ctx, result, out, stream);
    }
    if (groups > 1 && outer) {
        // option "group_device" = 0 (or "audit"): this thread waits for pass 0 and for every group - never inside a multi-GPU call, whose
        // rank threads wait with a deadline (hjgpu_phj_overlapped_async is hjgpu_phj_multi's / hjgpu_cpra_multi's local join)
        CHK(phj_grouped(ctx, groups, chunks, rk, rv, inner, sk, sv, outer, prm, out, stream));
    } else if (broadcast_applies(ctx->tune, inner, outer, chunks, prm)) {
        CHK(broadcast_enqueue(ctx, rk, rv, inner, sk, sv, outer, prm, out, stream, (hipEvent_t)inner_ready));
    } else {
        PhjPlan pl;
        CHK(phj_prepare(ctx, inner, outer, prm, chunks, &pl));
        CHK(phj_enqueue(ctx, pl, rk, rv, inner, sk, sv, outer, out, stream, (hipEvent_t)inner_ready));
    }
    if (d_result)
        HIPCHK(ctx, hj_copy_async(d_result, ctx->state.p, sizeof(hjgpu_result), stream));
    if (blocking) return finish_blocking(ctx, result, out, stream);
    return HJGPU_OK;
}

int hjgpu_phj(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner,
              const uint32_t *sk, const uint32_t *sv, size_t outer,
              const hjgpu_phj_params *prm, hjgpu_result *result, const hjgpu_output *out, void *stream)
{
    return phj_like(ctx, 1, rk, rv, inner, sk, sv, outer, prm, result, nullptr, out, stream, true);
}

int hjgpu_phj_async(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner,
                    const uint32_t *sk, const uint32_t *sv, size_t outer,
                    const hjgpu_phj_params *prm, hjgpu_result *d_result, void *stream)
{
    return phj_like(ctx, 1, rk, rv, inner, sk, sv, outer, prm, nullptr, d_result, nullptr, stream, false);
}

int hjgpu_phj_overlapped_async(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner,
                               const uint32_t *sk, const uint32_t *sv, size_t outer,
                               const hjgpu_phj_params *prm, hjgpu_result *d_result, void *stream,
                               void *inner_ready_event)
{
    return phj_like(ctx, 1, rk, rv, inner, sk, sv, outer, prm, nullptr, d_result, nullptr, stream, false,
                    inner_ready_event, true);
}

// ---- build side prepared once, probed by any number of batches ---------------------------------
int hjgpu_phj_build(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner, size_t max_outer,
                    const hjgpu_phj_params *prm, void *stream_)
{
    if (!ctx) return HJGPU_EINVAL;
    CHK(check_columns(ctx, rk, rv, inner));
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PhjPlan pl;
    CHK(refuse_capture(ctx, stream));
    CHK(phj_prepare(ctx, inner, max_outer, prm, 1, &pl));        // workspace and plan for the largest batch
    CHK(phj_enqueue(ctx, pl, rk, rv, inner, nullptr, nullptr, 0, nullptr, stream, nullptr, PHJ_BUILD_ONLY));
    memcpy(ctx->prepared_plan, &pl, sizeof(pl));
    ctx->prepared_inner = inner; ctx->prepared_max_outer = max_outer;
    ctx->prepared = true;
    return HJGPU_OK;
}

static int phj_probe_prepared(hjgpu_ctx *ctx, const uint32_t *sk, const uint32_t *sv, size_t outer,
                              hjgpu_result *result, hjgpu_result *d_result, const hjgpu_output *out,
                              void *stream_, bool blocking)
{
    if (!ctx) return HJGPU_EINVAL;
    if (!blocking) out = take_async_output(ctx, out);    // consumed by this call even if it fails below (see hjgpu_npj_async)
    PlainRows plain(ctx, blocking);                      // (settle() is this function's first statement)
    if (!ctx->prepared)
        return fail(ctx, HJGPU_EINVAL, "hjgpu_phj_probe: no prepared build side (hjgpu_phj_build), or another "
                                       "entry point has used the workspace since");
    if (outer > ctx->prepared_max_outer)
        return fail(ctx, HJGPU_EINVAL, "hjgpu_phj_probe: batch larger than the max_outer given to hjgpu_phj_build");
    CHK(check_columns(ctx, sk, sv, outer));
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PhjPlan pl;
    memcpy(&pl, ctx->prepared_plan, sizeof(pl));
    ctx->last_had_output = out && out->d_keys;
    // the build columns themselves are not read again: their partitions live in the workspace
    CHK(phj_enqueue(ctx, pl, nullptr, nullptr, ctx->prepared_inner, sk, sv, outer, out, stream, nullptr, PHJ_PROBE_ONLY));
    if (d_result)
        HIPCHK(ctx, hj_copy_async(d_result, ctx->state.p, sizeof(hjgpu_result), stream));
    if (blocking) return finish_blocking(ctx, result, out, stream);
    return HJGPU_OK;
}

int hjgpu_phj_probe(hjgpu_ctx *ctx, const uint32_t *sk, const uint32_t *sv, size_t outer,
                    hjgpu_result *result, const hjgpu_output *out, void *stream)
{
    return phj_probe_prepared(ctx, sk, sv, outer, result, nullptr, out, stream, true);
}

int hjgpu_phj_probe_async(hjgpu_ctx *ctx, const uint32_t *sk, const uint32_t *sv, size_t outer,
                          hjgpu_result *d_result, void *stream)
{
    return phj_probe_prepared(ctx, sk, sv, outer, nullptr, d_result, nullptr, stream, false);
}

int hjgpu_cpra(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner,
               const uint32_t *sk, const uint32_t *sv, size_t outer,
               const hjgpu_phj_params *prm, hjgpu_result *result, const hjgpu_output *out, void *stream)
{
    const uint32_t chunks = (prm && prm->chunks) ? prm->chunks : 8;
    return phj_like(ctx, chunks, rk, rv, inner, sk, sv, outer, prm, result, nullptr, out, stream, true);
}

int hjgpu_cpra_async(hjgpu_ctx *ctx, const uint32_t *rk, const uint32_t *rv, size_t inner,
                     const uint32_t *sk, const uint32_t *sv, size_t outer,
                     const hjgpu_phj_params *prm, hjgpu_result *d_result, void *stream)
{
    const uint32_t chunks = (prm && prm->chunks) ? prm->chunks : 8;
    return phj_like(ctx, chunks, rk, rv, inner, sk, sv, outer, prm, nullptr, d_result, nullptr, stream, false);
}

}  // extern "C"
