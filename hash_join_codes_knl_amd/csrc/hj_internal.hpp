// hj_internal.hpp — launch wrappers shared between the kernel files and the
// C-ABI layer (hjgpu_api.hip).  Everything here is enqueue-only on `stream`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hjgpu.h"

typedef unsigned long long u64;

// Geometry constants (also read by the planner on the host side).
// Scatter tile = block * vectors_per_thread * 4 tuples.  Pass 1 of the join pipeline
// (packed output) runs in whole-line mode when its carry buffers fit the LDS beside the
// tile, which decides the tile size from the fan-out; see hj_scatter_config().
constexpr uint32_t HJ_LINE_TUPLES = 16;          // packed (8-byte) tuples per 128-byte line
constexpr uint32_t HJ_STREAM_UNIT = 256;         // output slots one 16-lane group streams out at a time
constexpr uint32_t HJ_MAX_HEAVY = 128;           // listed units per tile: <= (16384 + 30 * 209) / HJ_STREAM_UNIT (carry + line offsets)
struct ScatterConfig { int block, vpt; bool carry; };
struct HjTuning;
ScatterConfig hj_scatter_config(const HjTuning &t, int pass, uint32_t F, bool out_packed);
int hj_scatter_tile(const HjTuning &t, int pass, uint32_t F, bool out_packed);
constexpr int HJ_JOIN_SLICE    = 1 << 16;       // probe tuples per work item (target)
// A build partition larger than one LDS table is joined in several table fills; the fills of such a
// partition are dealt to up to this many work items per probe slice (a heavy build key otherwise leaves
// hundreds of fills, each re-streaming the probe slice, to ONE workgroup).
constexpr int HJ_JOIN_FILL_GROUPS = 64;
// entries of the work-item directory: every partition has >= 1 probe slice, slices of ~HJ_JOIN_SLICE rows
inline size_t hj_join_items_capacity(size_t partitions, size_t outer_rows)
{
    return (partitions + outer_rows / HJ_JOIN_SLICE + 1) * HJ_JOIN_FILL_GROUPS;
}

// Join-kernel geometry: threads per workgroup, log2 of the LDS table slots, and
// probe vectors each lane keeps in flight.  Default 512 / 8192 slots (64 KiB) / 2;
// the "join_cfg" option ("block,log2slots,batch") selects another built variant (tuning).
struct JoinConfig {
    int block, log2slots, batch;
    int slots() const { return 1 << log2slots; }
    int cap() const { return slots() / 2; }      // max build tuples per table fill (load <= 0.5)
};
// 16 K-slot tables (128 KiB of LDS, one 1024-thread workgroup per CU): half the partitions for the same build
// side.  Chosen when the 8 K-slot tables would need more than HJGPU_MAX_PARTS partitions (|R| > ~114 M).
const JoinConfig &hj_join_config_big();
// is this geometry among the built instances of join_kernel (with a _UNIQUE instance, if asked for)?
bool hj_join_config_built(const JoinConfig &c, bool unique);

// Tuning / test switches of ONE context.  They are read from the environment once, in hjgpu_create
// (HJGPU_<NAME>), and can be set per context with hjgpu_set_option; nothing on a launch path looks at
// the environment or at mutable process-wide state, so contexts on different host threads (and on
// different devices) do not share anything.
struct HjTuning {
    int range_tiles = 0;            // "range_tiles": tiles per pass-1 range (0 = planned)
    bool dense2 = false;            // "dense2": dense final layout instead of line-aligned partitions
    bool npj_refhash = false;       // "npj_refhash": whole-join NPJ with the reference's bucket hash
    bool no_broadcast = false;      // "no_broadcast": no broadcast join for tiny build sides
    bool force_chained = false;     // "force_chained": chained fallback tables everywhere (tests)
    bool scatter_prof = false;      // "scatter_prof": K6 phase stamps (diagnostics; synchronises)
    bool unique = false;            // "unique": stop a probe at its first match (_UNIQUE, npj.cpp:288-290)
    bool piece_interleave = true;   // "piece_interleave": pass 2 over arrived pieces takes its tiles partition by partition (PlanArgs::seg_interleave)
    bool merged_plan = true;        // "merged_plan": whole joins on resident columns plan both relations with one set of K5 launches
    int placement = 12;             // "placement": candidate allocations tried for a large pass-1 twin (1 = take the first).  What makes a
                                    // block fast is its physical pages' spread over the memory channels (L2 tag / DRAM-credit stalls,
                                    // profiles/r04_placement_counters.txt): nothing to choose, only to measure.  A cap of 4 was tried in round 4
                                    // and left 3 of 12 fresh processes without a fast block (profiles/r04_placement_log.txt: 10 of 32
                                    // candidates are fast); the search stops at the first fast one, 3 candidates on average.
    int placement_ms = 500;         // "placement_ms": wall-clock budget of one placement search (0 = none): when it is spent the best block so far is
                                    // taken (round 4's driver run spent 3.07 s in a search that found no fast block among twelve)
    bool placement_log = false;     // "placement_log": the search prints every candidate's fill time and its choice to stderr (diagnostics)
    bool solo = false;              // "solo": the caller promises that NOTHING else runs on the device beside this context's blocking joins (one
                                    // process, one stream - the reference's programs): their partial-line stores (K6) stay plain,
                                    // 0.34 ms per 64 M x 1 G step faster.  Default 0: every store that may sit dirty in an L2 is non-temporal (round 5:
                                    // plain stores are lost beside other queues' kernel boundaries, 1.5 in 10^4 pipeline steps)
    bool audit = false;             // "audit" (diagnostics): every stage leaves a checksum of its output (audit_kernels.hip, hjgpu_audit_read)
    int hist_min_lds = 0;           // "hist_min_lds" (diagnostics): K4 asks for at least this many bytes of LDS per workgroup (nothing else then shares its CU)
    int reserve_cus = 0;            // "reserve_cus": CUs that K6's persistent grid leaves free (multi-GPU: room for RCCL's kernels)
    // "host_batch": probe rows per batch of the host calls.  -1 (default): the materialising call in batches of 64 Mi rows
    // (its rows go home behind the upload); the aggregate call uploads whole columns while they fit the device's free
    // memory (its device time is then that of ONE join over the whole probe side: 8.5 ms at 64 M x 1 G against 20 ms as the
    // sum of fifteen 64 Mi-row joins - the wall clock is the upload's either way) and falls back to batches when they do not.
    // 0: never batch; n > 0: always, n rows per batch.
    long long host_batch = -1;
    // Grouped plans (a third partitioning pass, hjgpu_api.hip phj_grouped): a build side of "group_from" tuples and more is
    // first split, with the probe side, into key-disjoint groups of about "group_inner" build tuples (64 M: 8 K-slot tables well
    // below their capacity; 1 G x 4 G: 68.1 ms against 72.4 at 100 M and 71.6 at 200 M, profiles/r04_grouped_sweep.txt), each joined by the
    // two-pass plan (the reference plans up to four passes from the partition count, phj.cpp:1791-1808).  group_from 0 = never.
    // "group_always" 1: every build side of group_from tuples and more (tests); 0: only where the cost estimate in
    // grouped_groups() says the extra pass pays (several table fills per partition AND a probe side large enough).
    long long group_from = 300000000, group_inner = 64000000;
    bool group_always = false;
    // "group_device" (default 1): a grouped plan is planned ON THE DEVICE - the groups' sizes and first rows stay in device memory
    // (group_desc_kernel), every kernel of a group's join reads them from there, and the whole plan is one stream-ordered sequence like
    // any other join: enqueue-only calls return at once, nobody waits for pass 0 (phj.cpp:1791-1863 plans and runs its passes inside
    // run_hj).  The workspace is sized for groups of up to (1 + "group_slack" / 100) x the mean group ("group_slack": per cent, default
    // 50); a group beyond that (heavy duplicates) is skipped and flagged, the result marked invalid, and the join is done again by the
    // HOST-planned form (option 0: always that form - the calling thread waits for pass 0 and for every group) at the caller's next
    // blocking touch point (the blocking calls themselves; hjgpu_get_async_status for the enqueue-only ones).
    bool group_device = true;
    int group_slack = 50;
    long long batch_tuples = 0;     // "batch_tuples": probe-side tuples per partitioning batch (0 = no batching, the default)
    JoinConfig join = {512, 13, 2}; // "join_cfg"
    int scatter_cfg[2][3] = {{0, 0, -1}, {0, 0, -1}};   // "scatter_cfg" / "scatter2_cfg": block, vpt, carry; block 0 = planned
};
void hj_tuning_from_env(HjTuning *t);
// returns false for an unknown name or a malformed value
bool hj_tuning_set(HjTuning *t, const char *name, const char *value);
inline const JoinConfig &hj_join_config_of(const HjTuning &t, bool big_tables) { return big_tables ? hj_join_config_big() : t.join; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: one flag per
// (kernel instance, device), so a process that drives several GPUs opts in on each of them.
struct HjPerDeviceOnce { unsigned char done[64]; };
int hj_allow_dynamic_lds(const void *kernel, int bytes, HjPerDeviceOnce *once);

// Geometry of partitioning pass 1, known on the host (sizes + alignment only):
// each chunk (segment) is cut into `ranges_per_chunk` contiguous ranges of whole
// tiles.  K4 counts per range, K5 turns the counts into per-range write bases,
// and K6 pass 1 walks the same ranges with private cursors - no global atomics.
constexpr uint32_t HJ_MAX_RANGE_ENTRIES = 1u << 24;   // ranges * F1 kept below this (64 MiB of counts)
// Independently partitioned chunks of a relation (CPRA: the reference's #threads, cpra2.cpp:1757-1827, 2023; its runs used 129
// threads and more - `./cpra 129 ...` is taken as asked -, the result does not depend on the number).  Beyond 8 chunks the plan is always two passes with
// line-aligned final partitions: the join then sees ONE region per partition whatever the number of chunks.
constexpr uint32_t HJ_MAX_CHUNKS = 256;
// work-claim counters inside MetaLayout::tickets (uint32 words, zeroed per join)
constexpr uint32_t HJ_TICKET_K4 = 0;                          // K4: [r * HJ_MAX_CHUNKS + chunk]
constexpr uint32_t HJ_TICKET_K6 = 2 * HJ_MAX_CHUNKS;          // K6: [2 * r + pass - 1]
constexpr uint32_t HJ_TICKET_MULTI_FILL = 2 * HJ_MAX_CHUNKS + 8;
constexpr uint32_t HJ_TICKET_JOIN = 2 * HJ_MAX_CHUNKS + 10;          // K7+K8: two uint64 work counters (8-byte aligned: the block starts on one), the
constexpr uint32_t HJ_TICKET_JOIN2 = 2 * HJ_MAX_CHUNKS + 12;         // second for the multi-fill half of a _UNIQUE join - zeroed with the tickets, per join
constexpr uint32_t HJ_TICKET_WORDS = 2 * HJ_MAX_CHUNKS + 16;
struct Pass1Geom {
    u64 n, part;                    // chunk c = rows [part * c, c + 1 == chunks ? n : part * (c + 1)): thread_beg / thread_end with
                                    // alignment 16 (npj.cpp:516-529; cpra2.cpp:1737-1742) - computed, not stored: any number of chunks
#if defined(__HIPCC__)
    __host__ __device__
#endif
    u64 beg(uint32_t c) const { return c >= chunks ? n : part * c; }
    uint32_t chunks;
    uint32_t align;                 // (address of the key column / 4) % 4
    uint32_t ranges_per_chunk;      // ceil(tiles of the largest chunk / tiles per range)
    uint32_t tile;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline u64 hj_tiles_of(u64 b, u64 e, uint32_t align, uint32_t tile)
{
    if (e <= b) return 0;
    const u64 gb = (align + b) & ~3ull, ge = align + e;
    return (ge - gb + tile - 1) / tile;
}

// One partitioning pass over `nseg` independent input segments.
struct ScatterArgs {
    const uint32_t *kin, *vin;      // input columns (in_packed: kin = packed tuples, vin unused)
    uint32_t *kout, *vout;          // output columns (out_packed: kout = packed tuples, vout unused)
    uint32_t in_packed, out_packed; // packed tuple = payload << 32 | key (8 bytes)
    const u64 *seg_off;             // [nseg+1] element offsets of the segments in kin/vin
    const u64 *tile_prefix;         // [nseg+1] exclusive prefix of tiles per segment
    const uint4 *tile_desc;         // pass 2: [tiles][2] {first row, end row of the segment | first slot of the tile, cursor row}
    u64 *cursors;                   // [nseg*F] pass 2: absolute output positions, advanced atomically;
                                    //          aligned_claims: lines claimed | tail tuples << 32
    const u64 *part_start, *part_end; // pass 2, aligned_claims: [nseg*F] first row / one past the last row of every partition
    uint32_t aligned_claims;
    uint32_t nseg, F, factor;
    uint32_t in_align;              // (address of kin / 4) % 4, same for vin
    // pass 1 only (ranged == 1): per-range bases instead of atomic cursors
    uint32_t ranged;
    uint32_t *work_counter;         // device, zeroed per launch: ticket of the next unclaimed range (pass 1) / tile (pass 2)
    Pass1Geom geom;
    uint32_t range_begin, range_count;   // pass 1: this launch covers ranges [range_begin, +range_count) (0, 0 = all)
    const u64 *range_base;          // [ranges][F] absolute output position of each (range, partition)
    u64 *prof;                      // diagnostics (HJGPU_SCATTER_PROF=1): s_memtime ticks per phase, else NULL
    uint32_t nt_partial;            // 1: the 8-byte (partial-line) stores are non-temporal too (every launch that is not solo: selects the
                                    // NTP instance of the kernel, see k6_store8)
    const u64 *dyn;                 // pass 1 of a device-planned group (else NULL): {first row, rows} of the input inside kin / vin, in device
                                    // memory - geom.n / geom.part are computed from it in the kernel (geom holds the CAPACITY's ranges and tile)
};

struct JoinArgs {
    const uint32_t *rk, *rv, *sk, *sv;   // co-partitioned columns (packed: rk / sk = packed tuples)
    uint32_t packed;                     // 1: payload << 32 | key tuples (the library's own passes)
    const u64 *roff, *soff;              // [chunks*P] first row of every partition, chunk-major
    const u64 *rend, *send;              // [chunks*P] one past its last row (dense layouts: roff + 1, soff + 1)
    const u64 *slice_prefix;             // [P+1] exclusive prefix of work items per partition
    const u64 *slices;                   // [P]   probe slices of partition q | fill groups << 32
    const uint32_t *item_part;           // [items] partition of work item w
    uint32_t P, chunks;
    uint32_t f1, F1, f2, F2;             // the passes that produced the partitions
    uint32_t p1_base;                    // pre-partitioned relations: the partitions are pass-1 partitions [p1_base, ...) of F1 (then
                                         // F1 = the exchange's total fan-out; the local partition of a key is
                                         // (H(key, f1, F1) - p1_base) * F2 + H(key, f2, F2))
    uint32_t tf0, tf1;                   // table hash / step multipliers
    uint32_t s_align;                    // (address of sk / 4) % 4
    hjgpu_result *result;                // device, accumulated atomically
    u64 *work_counter;                   // device, zeroed per launch: next unclaimed work item
    // materialised output (NULL keys = aggregate only)
    uint32_t *ok, *oov, *oiv;
    u64 block_size, block_limit;
    u64 *block_counter;                  // device
    u64 *final_offsets;                  // device [gridDim.x] end cursor per workgroup
    uint32_t *overflow;                  // device flag
    uint32_t nt_rows;                    // 1: result rows through non-temporal stores (every join but a solo one: selects the NTROWS instance)
    uint32_t big_tables;                 // hj_join_config_big() instead of hj_join_config()
    // broadcast join (tiny build side, nothing partitioned): P = 1, the relations are the caller's columns, the
    // empty sentinel is *sentinel (a value no build key equals, found by hj_launch_broadcast_meta)
    uint32_t broadcast;
    const uint32_t *sentinel;
    uint32_t force_chained;              // tests: skip the cuckoo fast path (option "force_chained")
    uint32_t unique;                     // _UNIQUE (npj.cpp:288-290): a probe key reports its first match only
    // a _UNIQUE join is two launches (join_kernel<.., UNIQUE, DEDUP>): the multi-fill half has its own work counter, goes on
    // in the open output blocks of the single-fill half (same grid, same final_offsets slots) and returns at once when the
    // plan counted no multi-fill partition (multi_fill, may be NULL = unknown)
    u64 *work_counter2;
    const uint32_t *multi_fill;
    uint32_t resume;                     // 1 (device-planned groups): every wave goes on in the output block its slot of final_offsets names (HJ_NO_CURSOR:
                                         // none yet) - the groups' joins share one block counter, one set of open blocks and ONE close_gaps at the end
};

struct PlanArgs {
    const u64 *counts[2];     // [chunks*P] fused two-level histograms of R (0) and S (1)
    u64 n[2];                 // relation sizes
    u64 chunk_beg[2][9];      // first row of every chunk (host-known: sizes only) - read when the chunks are pieces of any size
                              // (pre-partitioned relations, <= 8 pieces); regular chunks are computed:
    u64 *more[2] = {nullptr, nullptr};   // chunks > 8: [P] the counters of chunks 8 ... chunks - 1 added up (written by hj_launch_plan)
    uint32_t regular[2];      // 1: chunk c of relation r starts at row chunk_part[r] * c (Pass1Geom::part; up to HJ_MAX_CHUNKS chunks)
    u64 chunk_part[2];
    u64 *off2[2];             // [chunks*P + 1] first row of every final partition
    u64 *end2[2];             // [chunks*P] one past its last row
    uint32_t pad2;            // 1: final partitions start on 128-byte lines (two-pass plans)
    u64 *cur2[2];             // [chunks*P]
    u64 *off1[2];             // [chunks*F1 + 1]
    u64 *cur1[2];             // [chunks*F1]
    u64 *tp1[2];              // [chunks + 1] pass-1 tile prefix
    u64 *seg1[2];             // [chunks + 1] chunk boundaries
    u64 *tp2[2];              // [chunks*F1 + 1] pass-2 tile prefix
    uint4 *tdesc[2];          // [pass-2 tiles][2] per-tile descriptors (one independent load per tile in K6)
    uint32_t tdesc_cap;       // tiles the descriptor table holds
    u64 *slice_prefix;        // [P + 1]
    u64 *slices;              // [P]
    uint32_t *item_part;      // [hj_join_items_capacity] partition of every join work item
    uint32_t chunks, F1, F2;
    uint32_t in_align[2];     // alignment of the caller's input columns
    uint32_t tile1, tile2;    // tuples per tile in pass 1 / pass 2
    uint32_t slice;
    uint32_t cap;             // build rows per LDS table fill of the join kernel that will run (JoinConfig::cap)
    uint32_t mask;            // bit 0: plan R, bit 1: plan S, bit 2: join work items
    uint32_t unique;          // _UNIQUE joins: all table fills of a probe slice stay with ONE work item (see join_kernel)
    uint32_t *multi_fill = nullptr;   // += partitions with work whose build rows take more than one table fill (zeroed per join)
    // Chunked relations (one-GPU CPRA), line-aligned final layout: the pass-1 output is laid out PARTITION-major - the
    // chunks' regions of a pass-1 partition lie side by side, off1[c * F1 + p] still says where chunk c writes partition
    // p - so pass 2 sees F1 segments [seg2[p], seg2[p + 1]) exactly as after an unchunked pass 1, and pass 1 of every
    // chunk writes across the whole twin (chunk-major: 3.3 + 3.3 ms for the two passes at 2-8 chunks against 3.0 + 3.05)
    uint32_t p_major = 0;
    // Pieces that lie where they arrived (pre-partitioned relations, chunk-major in memory): pass 2 still takes its tiles
    // partition by partition - all pieces of pass-1 partition 0, then of partition 1 ... - so that every final partition
    // is written in one go (tile order only: entry i of the tile prefix is segment (i % chunks) * F1 + i / chunks)
    uint32_t seg_interleave = 0;
    u64 *seg2[2] = {nullptr, nullptr};   // [F1 + 1] p_major: the pass-1 partitions' bounds
    // a device-planned group (else NULL): {first row, rows} of relation r in device memory; n[r] and chunk_part[r] are then computed from it
    const u64 *dyn[2] = {nullptr, nullptr};
};

// A relation that arrives pass-1-partitioned in pieces (the multi-GPU CPRA's receiving side): piece c = rows
// [b[c], b[c + 1]) of one packed array, c < chunks <= 8 (one piece per source rank).
struct HjChunks {
    u64 b[9];
    uint32_t chunks;
};
int hj_launch_hist_packed(const u64 *tuples, const HjChunks &ch, uint32_t f1, uint32_t F1tot, uint32_t p1_base,
                          uint32_t F1, uint32_t f2, uint32_t F2, u64 *counts, int cus, hipStream_t stream);
int hj_launch_hist2(const uint32_t *keys, const Pass1Geom &geom,
                    uint32_t f1, uint32_t F1, uint32_t f2, uint32_t F2,
                    u64 *counts, uint32_t *range_counts, uint32_t *work_counter /* [chunks], zeroed */,
                    int cus, hipStream_t stream, size_t min_lds = 0, const u64 *dyn = nullptr /* device: {first row, rows}, see ScatterArgs::dyn */);
// Grouped plans on the device: from pass 0's offsets (dense prefixes of the F0 = G * bins counters of both relations) to one descriptor per
// group, desc[g] = {build first row, build rows, probe first row, probe rows} (first rows as pass 0 laid the groups out: hj_group_shift).
// A group with an empty side gets no rows at all (nothing can match); one beyond cap_r / cap_s rows gets none either and raises *skew.
int hj_launch_group_desc(const u64 *roff, const u64 *soff, uint32_t G, uint32_t bins, u64 cap_r, u64 cap_s, u64 n_r, u64 n_s, u64 *desc, uint32_t *skew,
                         hipStream_t stream);
// d_result of a device-planned grouped join: the aggregates, or all ones when a group was skipped (*skew != 0): never a plausible partial count
int hj_launch_group_result(const hjgpu_result *state, const uint32_t *skew, hjgpu_result *d_result, hipStream_t stream);
// own_count > 0 (chunks == 1): partitions [own_first, own_first + own_count) are laid out behind all others
// group_bins > 0 (chunks == 1, own_count == 0): groups of group_bins neighbouring partitions, each group on a 128-byte
// line: group g (whose dense start is row `dense`) starts at row dense + hj_group_shift(dense, g)
int hj_launch_range_base(const uint32_t *range_counts, const u64 *off1, u64 *range_base,
                         uint32_t chunks, uint32_t ranges_per_chunk, uint32_t F1, hipStream_t stream,
                         uint32_t own_first = 0, uint32_t own_count = 0, uint32_t group_bins = 0);
// 32 rows = one 128-byte line of a uint32 column.  Rounding the dense start up and adding one line per group keeps the
// groups apart whatever their sizes: start(g + 1) - start(g) >= rows(g) - 31 + 32.
__host__ __device__ inline u64 hj_group_shift(u64 dense_start, uint32_t g)
{
    return ((dense_start + 31) & ~(u64)31) - dense_start + (u64)32 * g;
}
int hj_launch_plan(const PlanArgs &a, hipStream_t stream);
// Batched partitioning of ONE relation (single chunk): the relation's pass-1 ranges are cut into batches of
// `ranges_per_batch`; pass 1 of a batch writes into a small REUSED buffer (dense layout starting at 0) and pass 2 of
// the batch reads it right away, while it is still in the 256 MiB Infinity Cache - the intermediate copy of the
// relation (16 of PHJ's 44 bytes per probe tuple) then mostly never reaches HBM.  This kernel derives, from K4's
// per-range counts: boff[b][F1 + 1] (the batch's pass-1 layout), range_base[range][F1] (write bases inside the batch
// buffer), tp2b[b][F1 + 1] (pass-2 tile prefix of the batch) and tdesc[b][cap][2] (its pass-2 tile descriptors).
struct BatchPlanArgs {
    const uint32_t *range_counts;   // [ranges][F1]
    u64 *range_base;                // [ranges][F1]
    u64 *boff;                      // [batches][F1 + 1]
    u64 *tp2b;                      // [batches][F1 + 1]
    uint4 *tdesc;                   // [batches][tdesc_cap][2]
    uint32_t tdesc_cap;
    uint32_t ranges, ranges_per_batch, F1, F2, tile2;
};
int hj_launch_batch_plan(const BatchPlanArgs &a, uint32_t batches, hipStream_t stream);
int hj_launch_scatter(const ScatterArgs &a, const HjTuning &t, int cus, hipStream_t stream);
int hj_launch_join(const JoinArgs &a, const HjTuning &t, int cus, hipStream_t stream);
int hj_launch_exscan(const u64 *in, u64 *out, uint32_t n, hipStream_t stream);
int hj_launch_offsets_to_counts(const u64 *off, u64 *counts, uint32_t P, hipStream_t stream);

// NPJ
int hj_launch_npj_build(const uint32_t *keys, const uint32_t *vals, size_t n, u64 *table,
                        size_t buckets, uint32_t factor, uint32_t *zero_key_flag,
                        int cus, hipStream_t stream, bool line_hash = false);
struct NpjProbeArgs {
    const uint32_t *keys, *vals;
    size_t n;
    const u64 *table;
    size_t buckets;
    uint32_t factor;
    uint32_t line_hash;                  // 1: walks start on 64-byte lines (the library's own tables)
    uint32_t unique;                     // _UNIQUE (npj.cpp:288-290): the walk ends at the key's first match
    hjgpu_result *result;
    uint32_t *ok, *oov, *oiv;
    u64 block_size, block_limit;
    u64 *block_counter;
    u64 *final_offsets;
    uint32_t *overflow;
};
int hj_launch_npj_probe(const NpjProbeArgs &a, int cus, hipStream_t stream, int *grid_out);

// K9: compact the per-wave partially filled tail blocks (npj.cpp:475-514).
// moves: scratch of 2*HJ_MAX_WORKERS entries of 24 bytes + HJ_MAX_WORKERS entries of 8 bytes.
constexpr uint32_t HJ_MAX_WORKERS = 8192;
int hj_launch_close_gaps_ex(uint32_t *k, uint32_t *ov, uint32_t *iv, const u64 *final_offsets,
                            uint32_t nworkers, u64 block_size, const u64 *block_counter,
                            const uint32_t *overflow, void *moves, uint32_t *nmoves,
                            u64 *dense_count, int cus, hipStream_t stream);
int hj_join_workers(const HjTuning &t, int cus, bool big_tables = false, bool unique = false);
// Metadata of a broadcast join, written on the device: one partition holding all of R ([0, inner)) and all of S
// ([0, outer)), `nslices` probe slices x `groups` fill groups (item_part must be zeroed by the caller), and a
// sentinel: a value whose low 14 bits no build key shares (inner <= 16383).
struct BroadcastMeta {
    u64 *roff, *rend, *soff, *send, *slice_prefix, *slices;
    uint32_t *sentinel;
};
int hj_launch_broadcast_meta(const uint32_t *inner_keys, size_t inner, size_t outer, uint32_t nslices,
                             uint32_t groups, const BroadcastMeta &m, hipStream_t stream);
int hj_npj_probe_grid(int cus, size_t n);

// option "audit" (audit_kernels.hip): read-only checks of a stage's output, on the stage's stream.  A record is
// HJ_AUDIT_STAGES x {misplaced tuples, sum of keys, sum of payloads, tuples}; a context keeps its last HJ_AUDIT_RING records.
// PHJ / CPRA calls: stage 0 probe side as read, 1 after pass 1, 2 final partitions; 3-5 the same for the build side (5 is
// checked again by every probe of a prepared build side); 6 the join's result (count, three sums); 7 {sequence number, kind
// (0 whole join, 1 build only, 2 probe only, 3 hjgpu_partition_packed*), build rows, probe rows}.  Partitioning calls: 0 input, 1 output.
constexpr int HJ_AUDIT_STAGES = 8, HJ_AUDIT_RING = 256;
struct HjAuditHash { uint32_t f1, F1, p1_base, f2, F2, modulo; };   // partition of a key: ((H(key, f1, F1) - p1_base) * F2 + H(key, f2, F2)); entry i holds partition i % modulo
int hj_audit_partitions(const u64 *tuples, const u64 *beg, const u64 *end /* NULL: beg + 1 */, uint32_t parts, const HjAuditHash &h,
                        u64 *rec, int cus, hipStream_t stream);
int hj_audit_sums_packed(const u64 *tuples, u64 b, u64 e, u64 *rec, int cus, hipStream_t stream);
int hj_audit_sums_columns(const uint32_t *k, const uint32_t *v, u64 n, u64 *rec, int cus, hipStream_t stream);
int hj_audit_own_last(const u64 *prefix, uint32_t F, uint32_t own_first, uint32_t own_count, u64 n, u64 *beg, u64 *end, hipStream_t stream);
int hj_audit_copy(const u64 *src, u64 *dst, uint32_t words, hipStream_t stream);
int hj_audit_meta(u64 *dst, u64 a, u64 b, u64 c, u64 d, hipStream_t stream);

// generator / checksums
int hj_launch_generate(u64 seed, size_t inner, size_t inner_begin, size_t inner_count,
                       size_t outer_total, size_t outer_begin,
                       size_t outer_count, uint32_t inner_factor, uint32_t outer_factor,
                       uint32_t *ik, uint32_t *iv, uint32_t *ok, uint32_t *ov, hipStream_t stream,
                       double zipf = 0.0, double selectivity = 1.0, u64 *d_expect = nullptr);
int hj_launch_stream_read(const void *p, size_t bytes, void *sink16, int cus, hipStream_t stream);
int hj_launch_copy_to_host(void *host_mapped, const void *d, size_t bytes, hipStream_t stream);
int hj_launch_row_sums(const u64 *counts, uint32_t F1, uint32_t F2, u64 *out, hipStream_t stream);
int hj_launch_random_line_read(const void *p, size_t bytes, size_t reads, void *sink16, int cus, hipStream_t stream);
int hj_launch_random_cas(void *p, size_t bytes, size_t ops, int in_flight, bool load_first, void *sink8, int cus, hipStream_t stream);
int hj_launch_fill_probe(void *p, size_t bytes, hipStream_t stream);
// the library's clears and device-to-device copies (gen_kernels.hip): own kernels with non-temporal stores instead of the runtime's
// hipMemsetAsync / hipMemcpyAsync, whose fill and copy kernels store plainly (the store policy: hj_device.hpp)
hipError_t hj_zero_async(void *p, size_t bytes, hipStream_t stream);
hipError_t hj_fill_async(void *p, uint32_t word, size_t bytes, hipStream_t stream);     // every 4-byte word = `word`
hipError_t hj_copy_async(void *dst, const void *src, size_t bytes, hipStream_t stream);
int hj_launch_column_sums(const uint32_t *keys, size_t n, uint32_t fa, uint32_t fb, u64 *sums3,
                          hipStream_t stream);
