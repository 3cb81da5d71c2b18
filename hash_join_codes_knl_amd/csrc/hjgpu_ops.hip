// hjgpu_ops.hip - the operator-level entry points of include/hjgpu.h (the reference's operators one by one: histogram /
// partition / build / probe, phj.cpp:693-1231, npj.cpp:190-364), the relations that arrive pass-1-partitioned (the receiving
// side of the multi-GPU CPRA), and the generator / measurement helpers.  Context, planning and whole joins: hjgpu_api.hip.
#include "hjgpu_ctx.hpp"

using namespace hjapi;

namespace hjapi {


// group_bins > 0: the partitions in groups of group_bins neighbours, every group on a 128-byte line (hj_group_shift; the
// columns then need room for n + 32 * (groups + 1) rows); d_offsets stay the dense prefix of the counts
int partition_columns(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                             uint32_t factor, uint32_t fanout, uint32_t group_bins, uint32_t *d_keys_out, uint32_t *d_vals_out,
                             uint64_t *d_offsets, void *stream_)
{
    if (!ctx || !d_offsets) return fail(ctx, HJGPU_EINVAL, "null pointer");
    if (fanout == 0 || fanout > HJGPU_MAX_FANOUT || !(factor & 1))
        return fail(ctx, HJGPU_EINVAL, "fanout must be in [1, 1024] and factor odd");
    if (n && (!d_keys_out || !d_vals_out)) return fail(ctx, HJGPU_EINVAL, "null output column");
    CHK(check_columns(ctx, d_keys, d_vals, n));
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(refuse_capture(ctx, stream));
    const Pass1Geom geom = make_geom(ctx->tune, d_keys, n, 1, fanout, false);
    MetaLayout sz = carve(nullptr, 1, fanout, fanout, geom.ranges_per_chunk);
    ctx->prepared = false;                 // the workspace is re-planned below
    CHK(ensure(ctx, ctx->meta, sz.total_bytes));
    MetaLayout m = carve(ctx->meta.p, 1, fanout, fanout, geom.ranges_per_chunk);
    // hjgpu_get_stats().ms_total afterwards = duration of the whole operator (histogram + plan + scatter)
    for (int i = 0; i < EV_COUNT; ++i) ctx->ev_valid[i] = false;
    ctx->last_algo = 2;
    record(ctx, EV_BEGIN, stream);
    HIPCHK(ctx, hj_zero_async(m.counts[0], m.counts_bytes, stream));
    if (n) CHK(hj_launch_hist2(d_keys, geom, factor, fanout, 1u, 1u, m.counts[0], m.range_counts[0], m.tickets,
                               ctx->cus, stream));
    PlanArgs pa;
    for (int r = 0; r < 2; ++r) {
        pa.counts[r] = m.counts[r]; pa.off2[r] = m.off2[r]; pa.end2[r] = m.end2[r]; pa.cur2[r] = m.cur2[r];
        pa.off1[r] = m.off1[r]; pa.cur1[r] = m.cur1[r]; pa.tp1[r] = m.tp1[r];
        pa.seg1[r] = m.seg1[r]; pa.tp2[r] = m.tp2[r];
    }
    pa.tdesc[0] = pa.tdesc[1] = nullptr; pa.tdesc_cap = 0; pa.pad2 = 0; pa.unique = 0;
    pa.n[0] = n; pa.n[1] = 0; pa.slice_prefix = m.slice_prefix; pa.slices = m.slices; pa.item_part = m.item_part;
    for (uint32_t c = 0; c < 9; ++c) { pa.chunk_beg[0][c] = c ? n : 0; pa.chunk_beg[1][c] = 0; }
    pa.regular[0] = pa.regular[1] = 0; pa.chunk_part[0] = pa.chunk_part[1] = 0;
    pa.chunks = 1; pa.F1 = fanout; pa.F2 = 1;
    pa.in_align[0] = align_of(d_keys); pa.in_align[1] = 0;
    pa.tile1 = pa.tile2 = geom.tile; pa.slice = HJ_JOIN_SLICE; pa.cap = (uint32_t)ctx->tune.join.cap(); pa.mask = 7u;
    CHK(hj_launch_plan(pa, stream));
    if (n) {
        CHK(hj_launch_range_base(m.range_counts[0], m.off1[0], m.range_base[0], 1,
                                 geom.ranges_per_chunk, fanout, stream, 0, 0, group_bins));
        ScatterArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.kin = d_keys; sa.vin = d_vals; sa.kout = d_keys_out; sa.vout = d_vals_out;
        sa.seg_off = m.seg1[0]; sa.tile_prefix = m.tp1[0]; sa.cursors = m.cur1[0];
        sa.nseg = 1; sa.F = fanout; sa.factor = factor; sa.in_align = align_of(d_keys);
        sa.ranged = 1; sa.work_counter = m.tickets + HJ_TICKET_K6; sa.geom = geom; sa.range_base = m.range_base[0];
        sa.in_packed = 0; sa.out_packed = 0;
        sa.nt_partial = ctx->rows_plain ? 0u : 1u;
                CHK(hj_launch_scatter(sa, ctx->tune, scatter_cus(ctx), stream));
    }
    HIPCHK(ctx, hj_copy_async(d_offsets, m.off2[0], ((size_t)fanout + 1) * sizeof(u64), stream));
    record(ctx, EV_GAPS, stream);
    return HJGPU_OK;
}

}  // namespace hjapi

extern "C" {

// ---- partition operators ------------------------------------------------------
int hjgpu_histogram(hjgpu_ctx *ctx, const uint32_t *d_keys, size_t n, uint32_t factor,
                    uint32_t fanout, uint64_t *d_counts, void *stream_)
{
    if (!ctx || !d_counts || (n && !d_keys)) return fail(ctx, HJGPU_EINVAL, "null pointer");
    if (fanout == 0 || fanout > HJGPU_MAX_PARTS || !(factor & 1))
        return fail(ctx, HJGPU_EINVAL, "fanout must be in [1, 32768] and factor odd");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hj_zero_async(d_counts, (size_t)fanout * sizeof(u64), stream));
    if (n) {
        // the per-range counts are a by-product here; they go to scratch
        const Pass1Geom g = make_geom(ctx->tune, d_keys, n, 1, 1, false);
        CHK(ensure(ctx, ctx->moves, ((size_t)g.ranges_per_chunk + 16) * sizeof(uint32_t)));
        uint32_t *ticket = (uint32_t *)ctx->moves.p + g.ranges_per_chunk;
        HIPCHK(ctx, hj_zero_async(ticket, 8 * sizeof(uint32_t), stream));
        CHK(hj_launch_hist2(d_keys, g, 1u, 1u, factor, fanout, (u64 *)d_counts,
                            (uint32_t *)ctx->moves.p, ticket, ctx->cus, stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(stream));
    return HJGPU_OK;
}

int hjgpu_partition_async(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                          uint32_t factor, uint32_t fanout, uint32_t *d_keys_out, uint32_t *d_vals_out,
                          uint64_t *d_offsets, void *stream_)
{
    return partition_columns(ctx, d_keys, d_vals, n, factor, fanout, 0, d_keys_out, d_vals_out, d_offsets, stream_);
}

int hjgpu_partition(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                    uint32_t factor, uint32_t fanout, uint32_t *d_keys_out, uint32_t *d_vals_out,
                    uint64_t *d_offsets, void *stream_)
{
    CHK(hjgpu_partition_async(ctx, d_keys, d_vals, n, factor, fanout, d_keys_out, d_vals_out, d_offsets, stream_));
    HIPCHK(ctx, hipStreamSynchronize((hipStream_t)stream_));
    return HJGPU_OK;
}

int hjgpu_join_partitions(hjgpu_ctx *ctx,
                          const uint32_t *rk, const uint32_t *rv, const uint64_t *roff,
                          const uint32_t *sk, const uint32_t *sv, const uint64_t *soff,
                          const hjgpu_phj_params *passes, hjgpu_result *result,
                          const hjgpu_output *out, void *stream_)
{
    if (!ctx || !passes || !roff || !soff || !rk || !rv || !sk || !sv)
        return fail(ctx, HJGPU_EINVAL, "null pointer");
    if (((uintptr_t)sk & 15) || ((uintptr_t)sv & 15))
        return fail(ctx, HJGPU_EALIGN, "probe columns must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PhjPlan pl;
    pl.C = 1;
    pl.F1 = passes->fanout1; pl.F2 = passes->fanout2 ? passes->fanout2 : 1;
    pl.P = pl.F1 * pl.F2;
    if (pl.F1 == 0 || pl.P < 2 || pl.P > HJGPU_MAX_PARTS) return fail(ctx, HJGPU_EINVAL, "fan-out out of range");
    pl.f1 = passes->factor1 ? passes->factor1 : DEFAULT_F1;
    pl.f2 = passes->factor2 ? passes->factor2 : DEFAULT_F2;
    pl.tf0 = passes->table_factor[0] ? passes->table_factor[0] : DEFAULT_TF0;
    pl.tf1 = passes->table_factor[1] ? passes->table_factor[1] : DEFAULT_TF1;
    if (!(pl.f1 & 1) || !(pl.f2 & 1) || !(pl.tf0 & 1) || !(pl.tf1 & 1))
        return fail(ctx, HJGPU_EINVAL, "hash factors must be odd");
    // the work-item directory is sized from the probe rows, which only the device knows here
    u64 s_ends[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(&s_ends[0], soff, sizeof(u64), hipMemcpyDeviceToHost, stream));
    HIPCHK(ctx, hipMemcpyAsync(&s_ends[1], soff + pl.P, sizeof(u64), hipMemcpyDeviceToHost, stream));
    HIPCHK(ctx, hipStreamSynchronize(stream));
    const size_t items_extra = hj_join_items_capacity(pl.P, (size_t)(s_ends[1] - s_ends[0])) - pl.P;
    MetaLayout sz = carve(nullptr, 1, pl.F1, pl.P, 1, items_extra);
    ctx->prepared = false;                 // the workspace is re-planned below
    CHK(ensure(ctx, ctx->meta, sz.total_bytes));
    CHK(ensure(ctx, ctx->state, sizeof(DevState)));
    MetaLayout m = carve(ctx->meta.p, 1, pl.F1, pl.P, 1, items_extra);
    DevState *st = reinterpret_cast<DevState *>(ctx->state.p);
    u64 bs = 0, bl = 0;
    const bool unique = ctx->tune.unique || (passes->flags & HJGPU_FLAG_UNIQUE);
    CHK(setup_output(ctx, out, (uint32_t)hj_join_workers(ctx->tune, ctx->cus, false, unique), &bs, &bl));
    record(ctx, EV_BEGIN, stream);
    HIPCHK(ctx, hj_zero_async(st, sizeof(DevState), stream));
    // counts = adjacent differences of the caller's offsets, then the usual plan
    // (re-derives identical offsets and the work-item prefix)
    CHK(hj_launch_offsets_to_counts((const u64 *)roff, m.counts[0], pl.P, stream));
    CHK(hj_launch_offsets_to_counts((const u64 *)soff, m.counts[1], pl.P, stream));
    PlanArgs pa;
    for (int r = 0; r < 2; ++r) {
        pa.counts[r] = m.counts[r]; pa.off2[r] = m.off2[r]; pa.end2[r] = m.end2[r]; pa.cur2[r] = m.cur2[r];
        pa.off1[r] = m.off1[r]; pa.cur1[r] = m.cur1[r]; pa.tp1[r] = m.tp1[r];
        pa.seg1[r] = m.seg1[r]; pa.tp2[r] = m.tp2[r];
    }
    pa.tdesc[0] = pa.tdesc[1] = nullptr; pa.tdesc_cap = 0; pa.pad2 = 0; pa.unique = unique ? 1u : 0u;
    pa.n[0] = pa.n[1] = 0; pa.slice_prefix = m.slice_prefix; pa.slices = m.slices; pa.item_part = m.item_part;
    for (int r = 0; r < 2; ++r) for (uint32_t c = 0; c < 9; ++c) pa.chunk_beg[r][c] = 0;
    pa.regular[0] = pa.regular[1] = 0; pa.chunk_part[0] = pa.chunk_part[1] = 0;
    pa.chunks = 1; pa.F1 = pl.F1; pa.F2 = pl.F2; pa.in_align[0] = pa.in_align[1] = 0;
    pa.tile1 = pa.tile2 = (uint32_t)hj_scatter_tile(ctx->tune, 2, 1, true); pa.slice = HJ_JOIN_SLICE;
    pa.cap = (uint32_t)ctx->tune.join.cap(); pa.mask = 7u;
    CHK(hj_launch_plan(pa, stream));
    for (int e : {EV_S_HIST, EV_S_PLAN, EV_S_SC1, EV_S_SC2, EV_WAITED, EV_R_HIST, EV_R_PLAN, EV_R_SC1, EV_R_SC2}) record(ctx, e, stream);
    JoinArgs ja;
    memset(&ja, 0, sizeof(ja));
    ja.rk = rk; ja.rv = rv; ja.sk = sk; ja.sv = sv;
    ja.roff = (const u64 *)roff; ja.soff = (const u64 *)soff;    // caller's offsets (may start at non-zero)
    ja.rend = ja.roff + 1; ja.send = ja.soff + 1;
    ja.slice_prefix = m.slice_prefix; ja.slices = m.slices; ja.item_part = m.item_part;
    ja.P = pl.P; ja.chunks = 1;
    ja.f1 = pl.f1; ja.F1 = pl.F1; ja.f2 = pl.f2; ja.F2 = pl.F2; ja.tf0 = pl.tf0; ja.tf1 = pl.tf1;
    ja.s_align = 0; ja.result = &st->result; ja.work_counter = &st->work_counter;
    ja.work_counter2 = &st->work_counter2;       // (multi_fill stays NULL: the caller's partitions were not counted)
    ja.unique = unique ? 1u : 0u;
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
                                    (const u64 *)ctx->final_offsets.p,
                                    (uint32_t)hj_join_workers(ctx->tune, ctx->cus, false, unique), bs, &st->block_counter,
                                    &st->overflow, ctx->moves.p, &st->nmoves, &st->dense, ctx->cus, stream));
    record(ctx, EV_GAPS, stream);
    ctx->last_algo = 1;
    return finish_blocking(ctx, result, out, stream);
}

// ---- NPJ operators ------------------------------------------------------------
int hjgpu_npj_build(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                    uint64_t *d_table, size_t buckets, uint32_t factor, void *stream_)
{
    if (!ctx || !d_table || (n && (!d_keys || !d_vals))) return fail(ctx, HJGPU_EINVAL, "null pointer");
    if (!(factor & 1) || buckets <= n) return fail(ctx, HJGPU_EINVAL, "factor must be odd and buckets > n");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->state, sizeof(DevState)));
    DevState *st = reinterpret_cast<DevState *>(ctx->state.p);
    HIPCHK(ctx, hj_zero_async(st, sizeof(DevState), stream));
    HIPCHK(ctx, hj_zero_async(d_table, buckets * sizeof(u64), stream));
    if (n) CHK(hj_launch_npj_build(d_keys, d_vals, n, (u64 *)d_table, buckets, factor, &st->zero_key, ctx->cus, stream));
    return finish_blocking(ctx, nullptr, nullptr, stream);
}

int hjgpu_npj_probe(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                    const uint64_t *d_table, size_t buckets, uint32_t factor,
                    hjgpu_result *result, const hjgpu_output *out, void *stream_)
{
    if (!ctx || !d_table || buckets == 0) return fail(ctx, HJGPU_EINVAL, "null pointer");
    CHK(check_columns(ctx, d_keys, d_vals, n));
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->state, sizeof(DevState)));
    HIPCHK(ctx, hj_zero_async(ctx->state.p, sizeof(DevState), stream));
    record(ctx, EV_BEGIN, stream); record(ctx, EV_R_HIST, stream);
    CHK(npj_probe_enqueue(ctx, d_keys, d_vals, n, (const u64 *)d_table, buckets, factor, out, stream, false, ctx->tune.unique));
    ctx->last_algo = 0;
    return finish_blocking(ctx, result, out, stream);
}

// ---- relations that arrive pass-1-partitioned (the receiving side of the multi-GPU CPRA) --------------------
static int partition_packed(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                            uint32_t factor, uint32_t fanout, uint32_t own_first, uint32_t own_count,
                            uint64_t *d_tuples_out, uint64_t *d_offsets, void *stream_,
                            uint32_t factor2 = 0, uint32_t fanout2 = 0, uint64_t *d_counts2 = nullptr)
{
    if (d_counts2 && (fanout2 < 1 || !(factor2 & 1) || factor2 == factor || (u64)fanout * fanout2 > HJGPU_MAX_PARTS))
        return fail(ctx, HJGPU_EINVAL, "fused counts: factor2 odd and different from factor, fanout * fanout2 <= 32768");
    if (!ctx || !d_offsets) return fail(ctx, HJGPU_EINVAL, "null pointer");
    if ((u64)own_first + own_count > fanout) return fail(ctx, HJGPU_EINVAL, "own_first + own_count must not exceed fanout");
    if (fanout == 0 || fanout > HJGPU_MAX_FANOUT || !(factor & 1))
        return fail(ctx, HJGPU_EINVAL, "fanout must be in [1, 1024] and factor odd");
    if (n && !d_tuples_out) return fail(ctx, HJGPU_EINVAL, "null output array");
    if ((uintptr_t)d_tuples_out & 127) return fail(ctx, HJGPU_EALIGN, "the packed output must be 128-byte aligned (whole-line writes)");
    CHK(check_columns(ctx, d_keys, d_vals, n));
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(refuse_capture(ctx, stream));
    const Pass1Geom geom = make_geom(ctx->tune, d_keys, n, 1, fanout, true);
    MetaLayout sz = carve(nullptr, 1, fanout, fanout, geom.ranges_per_chunk);
    ctx->prepared = false;                 // the workspace is re-planned below
    CHK(ensure(ctx, ctx->meta, sz.total_bytes));
    MetaLayout m = carve(ctx->meta.p, 1, fanout, fanout, geom.ranges_per_chunk);
    for (int i = 0; i < EV_COUNT; ++i) ctx->ev_valid[i] = false;
    ctx->last_algo = 2;                    // hjgpu_get_stats().ms_total = the whole operator
    record(ctx, EV_BEGIN, stream);
    u64 *audit = nullptr;                  // option "audit": stage 0 the columns as read, stage 1 the packed output where it lies
    CHK(audit_begin(ctx, 3, n, 0, stream, &audit));
    HIPCHK(ctx, hj_zero_async(m.counts[0], m.counts_bytes, stream));
    if (d_counts2) {
        // the same read of the keys also counts the RECEIVERS' second level (bin = p1 * fanout2 + p2, the join's fused
        // histogram): they then need no histogram pass of their own over what arrives (K4p).  The pass-1 counts of this
        // call are the row sums.
        HIPCHK(ctx, hj_zero_async(d_counts2, (size_t)fanout * fanout2 * sizeof(u64), stream));
        if (n) {
            u64 *fused = reinterpret_cast<u64 *>(d_counts2);
            CHK(hj_launch_hist2(d_keys, geom, factor, fanout, factor2, fanout2, fused, m.range_counts[0], m.tickets, ctx->cus, stream, (size_t)ctx->tune.hist_min_lds));
            CHK(hj_launch_row_sums(fused, fanout, fanout2, m.counts[0], stream));
        }
    } else if (n) CHK(hj_launch_hist2(d_keys, geom, factor, fanout, 1u, 1u, m.counts[0], m.range_counts[0], m.tickets, ctx->cus, stream));
    PlanArgs pa;
    for (int r = 0; r < 2; ++r) {
        pa.counts[r] = m.counts[r]; pa.off2[r] = m.off2[r]; pa.end2[r] = m.end2[r]; pa.cur2[r] = m.cur2[r];
        pa.off1[r] = m.off1[r]; pa.cur1[r] = m.cur1[r]; pa.tp1[r] = m.tp1[r];
        pa.seg1[r] = m.seg1[r]; pa.tp2[r] = m.tp2[r];
    }
    pa.tdesc[0] = pa.tdesc[1] = nullptr; pa.tdesc_cap = 0; pa.pad2 = 0; pa.unique = 0;
    pa.n[0] = n; pa.n[1] = 0; pa.slice_prefix = m.slice_prefix; pa.slices = m.slices; pa.item_part = m.item_part;
    for (uint32_t c = 0; c < 9; ++c) { pa.chunk_beg[0][c] = c ? n : 0; pa.chunk_beg[1][c] = 0; }
    pa.regular[0] = pa.regular[1] = 0; pa.chunk_part[0] = pa.chunk_part[1] = 0;
    pa.chunks = 1; pa.F1 = fanout; pa.F2 = 1;
    pa.in_align[0] = align_of(d_keys); pa.in_align[1] = 0;
    pa.tile1 = pa.tile2 = geom.tile; pa.slice = HJ_JOIN_SLICE; pa.cap = (uint32_t)ctx->tune.join.cap(); pa.mask = 1u;
    CHK(hj_launch_plan(pa, stream));
    if (n) {
        CHK(hj_launch_range_base(m.range_counts[0], m.off1[0], m.range_base[0], 1, geom.ranges_per_chunk, fanout, stream,
                                 own_first, own_count));
        ScatterArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.kin = d_keys; sa.vin = d_vals; sa.kout = reinterpret_cast<uint32_t *>(d_tuples_out); sa.vout = nullptr;
        sa.seg_off = m.seg1[0]; sa.tile_prefix = m.tp1[0]; sa.cursors = m.cur1[0];
        sa.nseg = 1; sa.F = fanout; sa.factor = factor; sa.in_align = align_of(d_keys);
        sa.ranged = 1; sa.work_counter = m.tickets + HJ_TICKET_K6; sa.geom = geom; sa.range_base = m.range_base[0];
        sa.in_packed = 0; sa.out_packed = 1;
        sa.nt_partial = ctx->rows_plain ? 0u : 1u;
                CHK(hj_launch_scatter(sa, ctx->tune, scatter_cus(ctx), stream));
    }
    HIPCHK(ctx, hj_copy_async(d_offsets, m.off2[0], ((size_t)fanout + 1) * sizeof(u64), stream));
    if (audit && n) {
        CHK(hj_audit_sums_columns(d_keys, d_vals, n, audit, ctx->cus, stream));
        CHK(ensure(ctx, ctx->audit_lay, (size_t)2 * (HJGPU_MAX_FANOUT + 1) * sizeof(u64)));
        u64 *beg = reinterpret_cast<u64 *>(ctx->audit_lay.p), *end = beg + HJGPU_MAX_FANOUT + 1;
        CHK(hj_audit_own_last(m.off2[0], fanout, own_first, own_count, n, beg, end, stream));
        const HjAuditHash h = {factor, fanout, 0u, 1u, 1u, fanout};
        CHK(audit_partitions(ctx, 1, reinterpret_cast<const u64 *>(d_tuples_out), beg, end, fanout, h, audit, stream));
    }
    record(ctx, EV_GAPS, stream);
    return HJGPU_OK;
}

int hjgpu_partition_packed_async(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                                 uint32_t factor, uint32_t fanout, uint64_t *d_tuples_out, uint64_t *d_offsets, void *stream)
{
    return partition_packed(ctx, d_keys, d_vals, n, factor, fanout, 0, 0, d_tuples_out, d_offsets, stream);
}

int hjgpu_partition_packed_own_last_async(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                                          uint32_t factor, uint32_t fanout, uint32_t own_first, uint32_t own_count,
                                          uint64_t *d_tuples_out, uint64_t *d_offsets, void *stream)
{
    return partition_packed(ctx, d_keys, d_vals, n, factor, fanout, own_first, own_count, d_tuples_out, d_offsets, stream);
}

int hjgpu_partition_packed_counted_async(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                                         uint32_t factor, uint32_t fanout, uint32_t own_first, uint32_t own_count,
                                         uint32_t factor2, uint32_t fanout2, uint64_t *d_tuples_out, uint64_t *d_offsets,
                                         uint64_t *d_counts2, void *stream)
{
    if (!d_counts2) return fail(ctx, HJGPU_EINVAL, "null counts array");
    return partition_packed(ctx, d_keys, d_vals, n, factor, fanout, own_first, own_count, d_tuples_out, d_offsets, stream,
                            factor2, fanout2, d_counts2);
}

// what hjgpu_phj_build_prepartitioned plans for a build side of `inner` rows in `fanout1` pass-1 partitions
static void prepartitioned_plan(const hjgpu_ctx *ctx, size_t inner, uint32_t k, const hjgpu_phj_params *prm, uint32_t *F2, bool *big)
{
    *big = false;
    double parts = ceil((double)inner / (ctx->tune.join.cap() * 0.85));
    if (parts > HJGPU_MAX_PARTS) { *big = true; parts = ceil((double)inner / (hj_join_config_big().cap() * 0.85)); }
    uint32_t f2 = (prm && prm->fanout2) ? prm->fanout2 : (uint32_t)std::max(2.0, ceil(parts / k));
    if (f2 < 2) f2 = 2;
    if (f2 > HJGPU_MAX_FANOUT) f2 = HJGPU_MAX_FANOUT;
    while ((u64)k * f2 > HJGPU_MAX_PARTS && f2 > 2) --f2;
    *F2 = f2;
}

int hjgpu_prepartitioned_plan(hjgpu_ctx *ctx, size_t inner, uint32_t fanout1, const hjgpu_phj_params *params,
                              uint32_t *fanout2, uint32_t *factor2)
{
    if (!ctx || !fanout1 || !fanout2 || !factor2) return HJGPU_EINVAL;
    bool big = false;
    prepartitioned_plan(ctx, inner, fanout1, params, fanout2, &big);
    *factor2 = (params && params->factor2) ? params->factor2 : DEFAULT_F2;
    return HJGPU_OK;
}

static int check_layout(hjgpu_ctx *ctx, const uint64_t *d_tuples, const hjgpu_prepartitioned *lay, HjChunks *ch, size_t *rows)
{
    if (!lay) return fail(ctx, HJGPU_EINVAL, "null layout");
    if (lay->chunks < 1 || lay->chunks > 8) return fail(ctx, HJGPU_EINVAL, "a pre-partitioned relation arrives in 1 to 8 pieces");
    if (!(lay->factor1 & 1) || lay->fanout1 == 0 || lay->fanout1_total > HJGPU_MAX_FANOUT ||
        (u64)lay->first_partition + lay->fanout1 > lay->fanout1_total)
        return fail(ctx, HJGPU_EINVAL, "pre-partitioned layout: odd factor1, fanout1 >= 1, first_partition + fanout1 <= fanout1_total <= 1024");
    ch->chunks = lay->chunks;
    for (uint32_t c = 0; c < 9; ++c) {
        ch->b[c] = lay->chunk_offsets[c <= lay->chunks ? c : lay->chunks];
        if (c && ch->b[c] < ch->b[c - 1]) return fail(ctx, HJGPU_EINVAL, "pre-partitioned layout: chunk_offsets must not decrease");
    }
    *rows = (size_t)(ch->b[lay->chunks] - ch->b[0]);
    if (*rows && !d_tuples) return fail(ctx, HJGPU_EINVAL, "null tuple array");
    if ((uintptr_t)d_tuples & 15) return fail(ctx, HJGPU_EALIGN, "packed tuples must be 16-byte aligned");
    return HJGPU_OK;
}

int hjgpu_phj_build_prepartitioned(hjgpu_ctx *ctx, const uint64_t *d_tuples, const hjgpu_prepartitioned *lay,
                                   size_t max_outer, const hjgpu_phj_params *prm, void *stream_)
{
    if (!ctx) return HJGPU_EINVAL;
    PrePieces pre;
    size_t inner = 0;
    CHK(check_layout(ctx, d_tuples, lay, &pre.ch[0], &inner));
    pre.tuples[0] = reinterpret_cast<const u64 *>(d_tuples);
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(refuse_capture(ctx, stream));
    // pass 1 is given (fan-out k = lay->fanout1 on this rank): pass 2 brings the build partitions down to one LDS table.
    // Always a second pass (F2 >= 2): the final partitions are then ONE line-aligned region each, whatever the number of pieces.
    hjgpu_phj_params p2;
    memset(&p2, 0, sizeof(p2));
    if (prm) p2 = *prm;
    const uint32_t k = lay->fanout1;
    bool big = false;
    uint32_t F2 = 0;
    prepartitioned_plan(ctx, inner, k, &p2, &F2, &big);
    p2.fanout1 = k; p2.fanout2 = F2;
    PhjPlan pl;
    CHK(phj_prepare(ctx, inner, max_outer, &p2, lay->chunks, &pl, true, big ? 1 : 0));
    pl.pre_f1 = lay->factor1; pl.pre_F1tot = lay->fanout1_total; pl.pre_base = lay->first_partition;
    if (pl.f2 == pl.pre_f1) return fail(ctx, HJGPU_EINVAL, "factor2 must differ from the exchange-level factor1 (same factor: the second pass would not split)");
    CHK(phj_enqueue(ctx, pl, nullptr, nullptr, inner, nullptr, nullptr, 0, nullptr, stream, nullptr, PHJ_BUILD_ONLY, &pre));
    memcpy(ctx->prepared_plan, &pl, sizeof(pl));
    ctx->prepared_inner = inner; ctx->prepared_max_outer = max_outer;
    ctx->prepared = true;
    return HJGPU_OK;
}

static int probe_prepartitioned(hjgpu_ctx *ctx, const uint64_t *d_tuples, const hjgpu_prepartitioned *lay, const uint64_t *d_counts,
                                hjgpu_result *d_result, void *stream_);

int hjgpu_phj_probe_prepartitioned_async(hjgpu_ctx *ctx, const uint64_t *d_tuples, const hjgpu_prepartitioned *lay,
                                         hjgpu_result *d_result, void *stream_)
{
    return probe_prepartitioned(ctx, d_tuples, lay, nullptr, d_result, stream_);
}

int hjgpu_phj_probe_prepartitioned_counted_async(hjgpu_ctx *ctx, const uint64_t *d_tuples, const hjgpu_prepartitioned *lay,
                                                 const uint64_t *d_counts, hjgpu_result *d_result, void *stream_)
{
    if (ctx && !d_counts) return fail(ctx, HJGPU_EINVAL, "null counts array");
    return probe_prepartitioned(ctx, d_tuples, lay, d_counts, d_result, stream_);
}

static int probe_prepartitioned(hjgpu_ctx *ctx, const uint64_t *d_tuples, const hjgpu_prepartitioned *lay, const uint64_t *d_counts,
                                hjgpu_result *d_result, void *stream_)
{
    if (!ctx) return HJGPU_EINVAL;
    const hjgpu_output *out = take_async_output(ctx, nullptr);   // consumed by this call even if it fails below (see hjgpu_npj_async)
    if (!ctx->prepared) return fail(ctx, HJGPU_EINVAL, "hjgpu_phj_probe_prepartitioned_async: no prepared build side, or another entry point has used the workspace since");
    PhjPlan pl;
    memcpy(&pl, ctx->prepared_plan, sizeof(pl));
    if (!pl.pre) return fail(ctx, HJGPU_EINVAL, "the prepared build side was not pre-partitioned (hjgpu_phj_build_prepartitioned)");
    PrePieces pre;
    size_t outer = 0;
    CHK(check_layout(ctx, d_tuples, lay, &pre.ch[1], &outer));
    if (lay->chunks != pl.C || lay->factor1 != pl.pre_f1 || lay->fanout1_total != pl.pre_F1tot ||
        lay->first_partition != pl.pre_base || lay->fanout1 != pl.F1)
        return fail(ctx, HJGPU_EINVAL, "the probe batch's layout differs from the prepared build side's (pieces, factor1, fan-outs, first partition)");
    if (outer > ctx->prepared_max_outer) return fail(ctx, HJGPU_EINVAL, "batch larger than the max_outer given to hjgpu_phj_build_prepartitioned");
    pre.tuples[1] = reinterpret_cast<const u64 *>(d_tuples);
    pre.counts[1] = reinterpret_cast<const u64 *>(d_counts);
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->last_had_output = out && out->d_keys;
    CHK(phj_enqueue(ctx, pl, nullptr, nullptr, ctx->prepared_inner, nullptr, nullptr, outer, out, stream, nullptr, PHJ_PROBE_ONLY, &pre));
    if (d_result)
        HIPCHK(ctx, hj_copy_async(d_result, ctx->state.p, sizeof(hjgpu_result), stream));
    return HJGPU_OK;
}

// ---- generator ------------------------------------------------------------------
int hjgpu_generate_range(hjgpu_ctx *ctx, uint64_t seed, size_t inner_total, size_t outer_total,
                         size_t inner_begin, size_t inner_count, size_t outer_begin, size_t outer_count,
                         uint32_t inner_factor, uint32_t outer_factor,
                         uint32_t *ik, uint32_t *iv, uint32_t *ok, uint32_t *ov, void *stream_)
{
    return hjgpu_generate_zipf(ctx, seed, inner_total, outer_total, inner_begin, inner_count, outer_begin,
                               outer_count, inner_factor, outer_factor, 0.0, ik, iv, ok, ov, stream_);
}

int hjgpu_generate_zipf(hjgpu_ctx *ctx, uint64_t seed, size_t inner_total, size_t outer_total,
                        size_t inner_begin, size_t inner_count, size_t outer_begin, size_t outer_count,
                        uint32_t inner_factor, uint32_t outer_factor, double zipf,
                        uint32_t *ik, uint32_t *iv, uint32_t *ok, uint32_t *ov, void *stream_)
{
    if (!ctx) return HJGPU_EINVAL;
    if (!(zipf >= 0.0) || zipf > 8.0) return fail(ctx, HJGPU_EINVAL, "zipf exponent must be in [0, 8]");
    if ((ik && !iv) || (ok && !ov)) return fail(ctx, HJGPU_EINVAL, "key column without payload column");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = hj_launch_generate(seed, inner_total, inner_begin, inner_count, outer_total, outer_begin,
                                outer_count, inner_factor, outer_factor, ik, iv, ok, ov, stream, zipf);
    if (rc != HJGPU_OK) return fail(ctx, rc, "generate: bad sizes or launch failure");
    HIPCHK(ctx, hipStreamSynchronize(stream));
    return HJGPU_OK;
}

int hjgpu_generate_select(hjgpu_ctx *ctx, uint64_t seed, size_t inner_total, size_t outer_total,
                          size_t inner_begin, size_t inner_count, size_t outer_begin, size_t outer_count,
                          uint32_t inner_factor, uint32_t outer_factor, double zipf, double selectivity,
                          uint32_t *ik, uint32_t *iv, uint32_t *ok, uint32_t *ov, hjgpu_result *expected, void *stream_)
{
    if (!ctx) return HJGPU_EINVAL;
    if (!(zipf >= 0.0) || zipf > 8.0) return fail(ctx, HJGPU_EINVAL, "zipf exponent must be in [0, 8]");
    if (!(selectivity >= 0.0) || selectivity > 1.0) return fail(ctx, HJGPU_EINVAL, "selectivity must be in [0, 1]");
    if ((ik && !iv) || (ok && !ov)) return fail(ctx, HJGPU_EINVAL, "key column without payload column");
    if (expected && outer_total < inner_total)
        return fail(ctx, HJGPU_EINVAL, "analytic aggregates need unique build keys (outer_total >= inner_total)");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    u64 *d_expect = nullptr;
    if (expected) {
        memset(expected, 0, sizeof(*expected));
        CHK(ensure(ctx, ctx->moves, 64));
        d_expect = (u64 *)ctx->moves.p;
        HIPCHK(ctx, hj_zero_async(d_expect, 4 * sizeof(u64), stream));
    }
    int rc = hj_launch_generate(seed, inner_total, inner_begin, inner_count, outer_total, outer_begin,
                                outer_count, inner_factor, outer_factor, ik, iv, ok, ov, stream, zipf, selectivity,
                                ok ? d_expect : nullptr);
    if (rc != HJGPU_OK) return fail(ctx, rc, "generate: bad sizes or launch failure");
    if (expected) HIPCHK(ctx, hipMemcpyAsync(expected, d_expect, sizeof(*expected), hipMemcpyDeviceToHost, stream));
    HIPCHK(ctx, hipStreamSynchronize(stream));
    return HJGPU_OK;
}

int hjgpu_generate(hjgpu_ctx *ctx, uint64_t seed, size_t inner, size_t outer_total,
                   size_t outer_begin, size_t outer_count, uint32_t inner_factor, uint32_t outer_factor,
                   uint32_t *ik, uint32_t *iv, uint32_t *ok, uint32_t *ov, void *stream_)
{
    return hjgpu_generate_range(ctx, seed, inner, outer_total, 0, inner, outer_begin, outer_count,
                                inner_factor, outer_factor, ik, iv, ok, ov, stream_);
}

int hjgpu_column_sums(hjgpu_ctx *ctx, const uint32_t *d_keys, size_t n, uint32_t fa, uint32_t fb,
                      uint64_t sums[3], void *stream_)
{
    if (!ctx || !sums || (n && !d_keys)) return HJGPU_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->moves, 64));
    u64 *d = (u64 *)ctx->moves.p;
    // hjgpu_get_stats().ms_total afterwards = duration of this one streaming read
    for (int i = 0; i < EV_COUNT; ++i) ctx->ev_valid[i] = false;
    ctx->last_algo = 2;
    record(ctx, EV_BEGIN, stream);
    CHK(hj_launch_column_sums(d_keys, n, fa, fb, d, stream));
    record(ctx, EV_GAPS, stream);
    HIPCHK(ctx, hipMemcpyAsync(sums, d, 3 * sizeof(u64), hipMemcpyDeviceToHost, stream));
    HIPCHK(ctx, hipStreamSynchronize(stream));
    return HJGPU_OK;
}

int hjgpu_stream_read_ms(hjgpu_ctx *ctx, const void *d_ptr, size_t bytes, float *ms, void *stream_)
{
    if (!ctx || !d_ptr || !ms) return HJGPU_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->moves, 64));
    for (int i = 0; i < EV_COUNT; ++i) ctx->ev_valid[i] = false;
    record(ctx, EV_BEGIN, stream);
    CHK(hj_launch_stream_read(d_ptr, bytes, ctx->moves.p, ctx->cus, stream));
    record(ctx, EV_GAPS, stream);
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[EV_GAPS]));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev[EV_BEGIN], ctx->ev[EV_GAPS]));
    ctx->last_algo = 2;
    return HJGPU_OK;
}

int hjgpu_random_line_read_ms(hjgpu_ctx *ctx, const void *d_ptr, size_t bytes, size_t reads, float *ms, void *stream_)
{
    if (!ctx || !d_ptr || !ms) return HJGPU_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->moves, 64));
    for (int i = 0; i < EV_COUNT; ++i) ctx->ev_valid[i] = false;
    record(ctx, EV_BEGIN, stream);
    CHK(hj_launch_random_line_read(d_ptr, bytes, reads, ctx->moves.p, ctx->cus, stream));
    record(ctx, EV_GAPS, stream);
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[EV_GAPS]));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev[EV_BEGIN], ctx->ev[EV_GAPS]));
    ctx->last_algo = 2;
    return HJGPU_OK;
}

int hjgpu_random_cas_ms(hjgpu_ctx *ctx, void *d_ptr, size_t bytes, size_t ops, int in_flight, int load_first, float *ms, void *stream_)
{
    if (!ctx || !d_ptr || !ms) return HJGPU_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CHK(ensure(ctx, ctx->moves, 64));
    HIPCHK(ctx, hj_zero_async(d_ptr, bytes, stream));              // every bucket empty: outside the timed span
    for (int i = 0; i < EV_COUNT; ++i) ctx->ev_valid[i] = false;
    record(ctx, EV_BEGIN, stream);
    CHK(hj_launch_random_cas(d_ptr, bytes, ops, in_flight, load_first != 0, ctx->moves.p, ctx->cus, stream));
    record(ctx, EV_GAPS, stream);
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[EV_GAPS]));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev[EV_BEGIN], ctx->ev[EV_GAPS]));
    ctx->last_algo = 2;
    return HJGPU_OK;
}

}  // extern "C"
