/*
 * hjgpu.h — C-ABI of libhjgpu: MI355X (gfx950) hash-join operators.
 *
 * This is the drop-in boundary for the hot path of xtcyclist/hash_join_codes_KNL.
 * The reference has no plugin/FFI layer: its "interface" is the operator
 * functions that run()/run_hj() call (npj.cpp:769-927, phj.cpp:1646-1949,
 * cpra2.cpp:1697-1986).  Each entry point below names the reference operator(s)
 * it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - Column layout is hj.h:1-72's: separate uint32 key and payload columns
 *     (inner = build side R, outer = probe side S).  `d_` pointers are DEVICE
 *     pointers (HBM); key and payload columns of one relation must be 16-byte
 *     aligned (the reference requires 64-byte alignment, npj.cpp:118-126).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     `*_async` entry points only enqueue; everything else returns after the
 *     work has completed.
 *   - Every function returns an HJGPU_* status (the reference aborts through
 *     assert(); a library reports).  hjgpu_last_error() has the detail text.
 *   - Hash function everywhere: H(key, f, N) = mulhi32((uint32)(key*f), N)
 *     (npj.cpp:200-201, phj.cpp:83-100, 721-722); factors must be odd.
 *   - Join result = order-free aggregates over the three output columns
 *     join_keys / join_outer_vals / join_inner_vals.
 *   - A context owns one workspace: calls on the SAME context must not overlap in
 *     time from several host threads, and joins enqueued on different streams through
 *     one context would share that workspace - use one context per stream / thread
 *     (contexts are cheap; several may live on one device).  Joins enqueued on ONE
 *     stream may be queued back to back without host synchronisation.
 */
#ifndef HJGPU_H
#define HJGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HJGPU_OK          0
#define HJGPU_EINVAL      1   /* null pointer, even factor, fan-out out of range ...        */
#define HJGPU_EALIGN      2   /* column pointer not 16-byte aligned                         */
#define HJGPU_ENOMEM      3   /* device allocation failed                                   */
#define HJGPU_EHIP        4   /* HIP runtime error (text in hjgpu_last_error)               */
#define HJGPU_EZEROKEY    5   /* NPJ only: key 0 is the empty-bucket sentinel (npj.cpp:583) */
#define HJGPU_EOVERFLOW   6   /* materialised output exceeded its capacity
                                 (reference: assert(o <= block_limit), npj.cpp:245)          */
#define HJGPU_ENODEVICE   7   /* no gfx950-class device visible                             */

/* params->flags */
#define HJGPU_FLAG_UNIQUE 1u      /* the reference's -D_UNIQUE build (npj.cpp:288-290, 436-438; phj.cpp:459,
                                     635; cpra2.cpp:456, 625, 698): a probe tuple reports its FIRST match only
                                     and its walk ends there.  With unique build keys the result is the same
                                     as without the flag; with duplicate build keys count / sum_keys /
                                     sum_outer_vals count every probe tuple that has a match once, and
                                     sum_inner_vals adds the payload of ONE of its build tuples (which one
                                     depends on insertion order, in the reference as here).               */

#define HJGPU_MAX_FANOUT  1024u   /* per partitioning pass                                   */
#define HJGPU_MAX_PARTS   32768u  /* fanout1 * fanout2                                       */

typedef struct hjgpu_ctx hjgpu_ctx;

typedef struct {
    uint64_t count;            /* J                                                          */
    uint64_t sum_keys;         /* sum join_keys[j]        mod 2^64                           */
    uint64_t sum_outer_vals;   /* sum join_outer_vals[j]  (probe-side payload)               */
    uint64_t sum_inner_vals;   /* sum join_inner_vals[j]  (build-side payload)               */
} hjgpu_result;

/* Optional materialised output: three device columns of `capacity` uint32.
 * Blocks of block_size tuples are claimed with one atomic each, exactly the
 * reference's protocol (npj.cpp:244-246, 312-316); after the join the dense
 * prefix [0, count) holds the result (close_gaps, npj.cpp:475-514). */
typedef struct {
    uint32_t *d_keys;
    uint32_t *d_outer_vals;
    uint32_t *d_inner_vals;
    size_t    capacity;        /* in tuples; multiple of block_size                          */
    size_t    block_size;      /* power of two; 0 = 65536 (npj.cpp:945)                      */
} hjgpu_output;

typedef struct {
    uint32_t fanout1, fanout2; /* pass fan-outs; 0 = choose from |R| and the LDS table size  */
    uint32_t factor1, factor2; /* odd pass multipliers; 0 = library defaults                 */
    uint32_t table_factor[2];  /* odd table hash / step multipliers; 0 = defaults            */
    uint32_t chunks;           /* CPRA only: number of independently partitioned chunks
                                  (the reference's #threads, cpra2.cpp:1757-1827, 2023); 0 = 8;
                                  1..256 (HJGPU_EINVAL beyond: the result does not depend on it;
                                  beyond 8 the plan is always two passes)                    */
    uint32_t flags;            /* HJGPU_FLAG_*                                               */
} hjgpu_phj_params;

typedef struct {
    double   load;             /* buckets = inner/load (npj.cpp:944-947); 0 = 0.25           */
    uint32_t factor;           /* odd multiplier; 0 = default                                */
    uint32_t flags;            /* HJGPU_FLAG_*                                               */
} hjgpu_npj_params;

/* Per-phase device times of the last join on this context (hipEvent based). */
typedef struct {
    float    ms_total;
    float    ms_histogram;     /* K4: fused two-level histogram, R and S                     */
    float    ms_plan;          /* K5: prefix sums / cursors                                  */
    float    ms_scatter1;      /* K6 pass 1, R and S                                         */
    float    ms_scatter2;      /* K6 pass 2, R and S                                         */
    float    ms_join;          /* K7+K8 (PHJ/CPRA) or K3 probe (NPJ)                         */
    float    ms_build;         /* NPJ: K1 clear + K2 build                                   */
    float    ms_close_gaps;    /* K9                                                         */
    float    ms_inner_wait;    /* stream time spent waiting for the build side to arrive
                                  (hjgpu_phj_overlapped_async), 0 otherwise                  */
    float    ms_upload;        /* hjgpu_join_host: host columns -> HBM (wall clock, pipelined
                                  with the probe side's partitioning for PHJ / CPRA)         */
    float    ms_download;      /* hjgpu_join_host_rows: result columns -> host (wall clock)  */
    float    ms_reserve;       /* wall clock this context has spent growing its workspace so far (hjgpu_reserve / the
                                  first join of a size): allocations plus the placement search for the probe side's
                                  pass-1 twin (option "placement": up to 12 candidate blocks held and filled twice);
                                  never inside a timed join once the workspace is reserved                   */
    uint32_t fanout1, fanout2; /* what was used                                              */
    uint32_t batches;          /* PHJ: batches the probe side was partitioned in (both passes of a batch
                                  back to back, its intermediate copy kept in the Infinity Cache; their
                                  time is reported as ms_scatter1, ms_scatter2 = 0); 0 = unbatched    */
    uint64_t buckets;          /* NPJ table size                                             */
    float    ms_scatter0;      /* grouped plans (a build side of option "group_from" tuples and more, the reference's
                                  third pass, phj.cpp:1791-1808): pass 0, both relations split into `groups`
                                  key-disjoint groups; every other phase is then the SUM over the groups' joins,
                                  ms_total the sum of all of it                                              */
    uint32_t groups;           /* 0 = the plain two-pass plan                                */
    /* the context's last placement search (option "placement": candidate blocks for a buffer of 1 GiB and more that K6's
     * pass 1 scatters into; outside every timed join): blocks allocated and filled, 1 when the search's wall-clock budget
     * (option "placement_ms", default 500) ended it before a fast block was found, the kept block's fill time and size
     * (bytes / ms = its fill rate: >= 5.5 TB/s is the fast kind), and what the search cost: its wall clock
     * beyond the kept block's own hipMalloc (ms_reserve above is ALL growth of the workspace: every allocation, and in a fresh
     * process the first kernel's code-object load) */
    uint32_t placement_tried, placement_timeboxed;
    float    placement_fill_ms;
    float    placement_search_ms;
    uint64_t placement_bytes;
} hjgpu_stats;

typedef struct {
    char     name[128];
    char     arch[64];
    int      compute_units;
    int      lds_bytes_per_block;
    uint64_t hbm_bytes;
} hjgpu_device_info;

/* ---- context ---------------------------------------------------------------- */
/* hash of the kernel sources this library was built from: measurements (profiles/ traffic files) name the
 * kernels they were taken with, bench.py refuses counters of other kernels */
const char *hjgpu_kernel_hash(void);
/* hash of EVERY source of this library (every .hip and .hpp file under csrc, this header): stress / validation logs and the bench line name the
 * library they were taken with - a change of the orchestration changes it, the kernel hash above need not */
const char *hjgpu_library_hash(void);
int  hjgpu_device_count(int *count);                      /* visible GPUs (hosts that do not link HIP) */
int  hjgpu_create(int device, hjgpu_ctx **ctx);           /* device < 0: current device      */
int  hjgpu_destroy(hjgpu_ctx *ctx);
const char *hjgpu_last_error(const hjgpu_ctx *ctx);
const char *hjgpu_status_string(int status);
int  hjgpu_get_device_info(hjgpu_ctx *ctx, hjgpu_device_info *info);
/* Tuning / test switches of ONE context (the reference's compile-time macros and hard-coded constants,
 * SURVEY.md section 5 "Config / flags").  hjgpu_create reads HJGPU_<NAME> from the environment once;
 * nothing else in the library looks at the environment, and nothing on a launch path reads mutable
 * process-wide state.  Names: "unique" (as HJGPU_FLAG_UNIQUE, for every join of the context,
 * including the operator-level hjgpu_npj_probe), "force_chained", "no_broadcast", "dense2", "npj_refhash",
 * "scatter_prof", "merged_plan", "piece_interleave" (0 / 1); "range_tiles" (n); "join_cfg" ("block,log2slots,batch"); "scatter_cfg" /
 * "scatter2_cfg" ("block,vectors[,carry]"); "placement" (candidate allocations for the probe side's pass-1
 * twin, 1..16; "placement_ms": the search's wall-clock budget, default 500, 0 = none; "placement_log" 1: every candidate's fill time on stderr); "batch_tuples" (n, 0 = off); "group_from" / "group_inner" (tuples), "group_always" and "group_device" (0 / 1), "group_slack" (per cent): the
 * grouped plans of hjgpu_phj / hjgpu_cpra (below); "solo" (0 / 1, default 0: the caller promises that nothing else runs on the device beside this context's BLOCKING joins - one process,
 * one stream, as the reference's programs: the joins' partial-line and row stores then stay plain, 4 % faster; without the
 * promise every store that could sit dirty in an L2 is non-temporal, because plain stores ARE lost beside other queues' kernel
 * boundaries: 1.5 in 10^4 steps of the multi-GPU pipeline, DESIGN section 3 "Round 5"); diagnostics: "audit" (0 / 1: every stage of a join leaves a checksum of its output,
 * hjgpu_audit_read below), "hist_min_lds" (bytes of LDS a histogram workgroup asks for at least: nothing else then shares its CU).
 * Unknown names and malformed values: HJGPU_EINVAL. */
int  hjgpu_set_option(hjgpu_ctx *ctx, const char *name, const char *value);
/* Option "audit" (diagnostics; the reference's workers meet at barriers between their phases, phj.cpp:1715-1770,
 * cpra2.cpp:1834-1840 - here the phases are kernels on several streams, and when a step of a pipeline comes out wrong the
 * aggregates do not say which kernel's output lacked the tuples): every PHJ / CPRA / partitioning call of the context is
 * followed, stage by stage and on the call's stream, by read-only kernels that check the stage's output where it lies.
 * A record is 8 stages x {tuples lying in a partition their key does not hash to, sum of keys, sum of payloads, tuples}:
 * 0 probe side as read, 1 after pass 1, 2 final partitions; 3-5 the same for the build side (5 again at every probe of a
 * prepared build side); 6 the join's result; 7 {sequence number, kind (0 whole join, 1 build only, 2 probe only,
 * 3 hjgpu_partition_packed*), build rows, probe rows}.  The context keeps its last 256 records.  *next_seq (may be NULL) =
 * the sequence number the next call will get; records[count][32] = the calls first_seq .. first_seq + count - 1 (waits for
 * `stream`). */
int  hjgpu_audit_read(hjgpu_ctx *ctx, uint64_t *next_seq, uint64_t first_seq, uint32_t count, uint64_t *records, void *stream);
/* Option "audit", second look (diagnostics): the partition checks of the context's LAST audited call done again with the device quiet
 * (hipDeviceSynchronize first) - by a fresh kernel, and on the host from a copy of the partitions made with hipMemcpy (the copy
 * engine reads memory, not an XCD's L2).  words[checks][9] = {stage, the four words as the fresh kernel counts them, the four words
 * as the host counts them}; *checks = the checks there are, nothing is done when capacity (in checks) is smaller.  A stage whose
 * record was wrong and whose memory is wrong here lost stores; one whose memory is right here was read stale
 * (tools/stress_cpra.py --forensics prints the verdict for every wrong step). */
int  hjgpu_audit_recheck(hjgpu_ctx *ctx, uint64_t *words, size_t capacity, size_t *checks);
/* Pre-size the internal workspace (partition scratch twins = hj.h's [1]
 * columns, NPJ table) so that no allocation happens inside a timed join. */
int  hjgpu_reserve(hjgpu_ctx *ctx, size_t inner_tuples, size_t outer_tuples);
int  hjgpu_get_stats(hjgpu_ctx *ctx, hjgpu_stats *stats);

/* ---- device memory helpers for hosts that do not link HIP (mamalloc/free,
 * npj.cpp:118-126, and the fread targets npj.cpp:1013-1039) ------------------- */
int  hjgpu_malloc(hjgpu_ctx *ctx, void **d_ptr, size_t bytes);
/* hjgpu_malloc with the placement search of the library's own workspace (option "placement": up to 12 candidate blocks
 * are held and filled, the fastest is kept; buffers below 1 GiB: plain hjgpu_malloc): for buffers of a gigabyte and more
 * that a join writes into at many places at once - the result columns of a materialising join (its time differs by up
 * to 13 % between allocations of the same columns).  The search's wall clock is added to hjgpu_stats.ms_reserve. */
int  hjgpu_malloc_placed(hjgpu_ctx *ctx, void **d_ptr, size_t bytes);
int  hjgpu_free(hjgpu_ctx *ctx, void *d_ptr);
int  hjgpu_memcpy_h2d(hjgpu_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int  hjgpu_memcpy_d2h(hjgpu_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int  hjgpu_synchronize(hjgpu_ctx *ctx, void *stream);
/* Page-locked host memory for the fread() targets of the host programs (the reference's
 * mamalloc'd columns, npj.cpp:982-1000): hjgpu_join_host DMAs such columns straight to HBM at
 * the PCIe rate; pageable columns are staged through two pinned buffers instead. */
int  hjgpu_host_alloc(hjgpu_ctx *ctx, void **h_ptr, size_t bytes);
int  hjgpu_host_free(hjgpu_ctx *ctx, void *h_ptr);

/* ---- partition operators ------------------------------------------------------ */
/* histogram(), phj.cpp:693 (shared form 773): d_counts[p] = |{i: H(key_i,f,F)=p}|. */
int  hjgpu_histogram(hjgpu_ctx *ctx, const uint32_t *d_keys, size_t n,
                     uint32_t factor, uint32_t fanout, uint64_t *d_counts, void *stream);
/* histogram() + interleave() + partition() (phj.cpp:693, 1263, 1029): permutes
 * (key, payload) so that partition p occupies [d_offsets[p], d_offsets[p+1]);
 * order inside a partition is unspecified, as in the reference.
 * d_offsets: fanout+1 uint64 (64-bit: SURVEY F10). */
int  hjgpu_partition(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                     uint32_t factor, uint32_t fanout,
                     uint32_t *d_keys_out, uint32_t *d_vals_out, uint64_t *d_offsets,
                     void *stream);
/* Enqueue-only form (no host synchronisation; d_offsets is valid once `stream` has reached this point):
 * the exchange-level partitioning of the multi-GPU CPRA keeps several GPUs busy from one host thread. */
int  hjgpu_partition_async(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                           uint32_t factor, uint32_t fanout,
                           uint32_t *d_keys_out, uint32_t *d_vals_out, uint64_t *d_offsets,
                           void *stream);
/* build()+probe() over co-partitioned relations (phj.cpp:307, 399; the commented
 * join loop phj.cpp:1869-1924 / live cpra2.cpp:1883-1971): for every partition p,
 * R rows [d_inner_offsets[p], [p+1]) are loaded into an LDS hash table and S rows
 * [d_outer_offsets[p], [p+1]) probe it.  (factor1, fanout1, factor2, fanout2)
 * must be the passes that produced the partitions (needed to pick a per-partition
 * empty sentinel, phj.cpp:1886-1897; fanout2 = 1 for a single pass). */
int  hjgpu_join_partitions(hjgpu_ctx *ctx,
                           const uint32_t *d_inner_keys, const uint32_t *d_inner_vals,
                           const uint64_t *d_inner_offsets,
                           const uint32_t *d_outer_keys, const uint32_t *d_outer_vals,
                           const uint64_t *d_outer_offsets,
                           const hjgpu_phj_params *passes,
                           hjgpu_result *result, const hjgpu_output *out, void *stream);

/* ---- NPJ operators ------------------------------------------------------------- */
/* set()+build(), npj.cpp:366, 190: d_table = uint64[buckets] of (val<<32)|key. */
int  hjgpu_npj_build(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                     uint64_t *d_table, size_t buckets, uint32_t factor, void *stream);
/* probe()+close_gaps(), npj.cpp:216, 475. */
int  hjgpu_npj_probe(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                     const uint64_t *d_table, size_t buckets, uint32_t factor,
                     hjgpu_result *result, const hjgpu_output *out, void *stream);

/* ---- whole joins on HBM-resident columns (replace run()/run_hj()) ---------------- */
/* run(), npj.cpp:769-927 */
int  hjgpu_npj(hjgpu_ctx *ctx,
               const uint32_t *d_inner_keys, const uint32_t *d_inner_vals, size_t inner,
               const uint32_t *d_outer_keys, const uint32_t *d_outer_vals, size_t outer,
               const hjgpu_npj_params *params,
               hjgpu_result *result, const hjgpu_output *out, void *stream);
/* run_hj(), phj.cpp:1646-1949.  Passes (phj.cpp:1791-1808 plans 1-4 of equal fan-out from the partition count): one
 * or two passes up to HJGPU_MAX_PARTS partitions; a build side beyond their reach (~228 M tuples) whose probe side is
 * large enough for a further pass to pay is joined by a GROUPED plan - pass 0 splits both relations into key-disjoint
 * groups of about "group_inner" (64 M) build tuples, each joined by the two-pass plan, aggregates and rows added up
 * (hjgpu_stats.groups / ms_scatter0).  The groups are planned ON THE DEVICE (round 6; option "group_device", default 1): pass 0 leaves
 * the groups' offsets in device memory, one small kernel turns them into a descriptor per group (first row and rows of its build and probe
 * columns), and every kernel of a group's join reads its geometry from there - the whole plan is ONE stream-ordered sequence on the
 * caller's stream, exactly like an ungrouped join (phj.cpp:1791-1863 plans and runs its passes inside run_hj).  The *_async forms
 * therefore return at once for grouped plans too, hold no stream in hardware and use no thread of their own.  The workspace of a
 * group's join is planned for up to (1 + "group_slack" / 100) x the mean group (default 50); a group beyond that - heavy duplicates -
 * is SKIPPED and flagged on the device, and the join is then not valid: an *_async call's d_result holds all ones (never a plausible
 * partial count), and the join is done again in the host-planned form (the calling thread waits for pass 0 and plans every group's
 * join from its size) at the caller's next blocking touch point - inside the call for the blocking forms, in hjgpu_get_async_status for
 * the enqueue-only ones (which must therefore be called before d_result or the rows of a grouped plan are trusted; it costs a stream
 * synchronisation when nothing was skipped).  Option "group_device" = 0: always the host-planned form (the *_async forms then wait
 * inside the call); option "audit" implies it.  hjgpu_phj_overlapped_async (the local join of hjgpu_phj_multi / hjgpu_cpra_multi) is
 * always planned on the device; the rank's thread asks for the status after a wait with the communicator's deadline.  Explicit fan-outs in
 * params are never grouped; option "group_from" = 0 turns the plan off. */
int  hjgpu_phj(hjgpu_ctx *ctx,
               const uint32_t *d_inner_keys, const uint32_t *d_inner_vals, size_t inner,
               const uint32_t *d_outer_keys, const uint32_t *d_outer_vals, size_t outer,
               const hjgpu_phj_params *params,
               hjgpu_result *result, const hjgpu_output *out, void *stream);
/* run_hj(), cpra2.cpp:1697-1986 (chunked partitioning, owner gathers slices) */
int  hjgpu_cpra(hjgpu_ctx *ctx,
                const uint32_t *d_inner_keys, const uint32_t *d_inner_vals, size_t inner,
                const uint32_t *d_outer_keys, const uint32_t *d_outer_vals, size_t outer,
                const hjgpu_phj_params *params,
                hjgpu_result *result, const hjgpu_output *out, void *stream);
/* hjgpu_npj_async cannot report HJGPU_EZEROKEY (a build key of 0 is skipped by the build, npj.cpp:583, and the
 * blocking hjgpu_npj says so): callers of the enqueue-only form must keep key 0 out of the build side themselves.
 * Enqueue-only forms: the aggregates land in d_result (device memory, 32 bytes);
 * no host synchronisation, so a caller can time with its own events and keep several
 * joins in flight on one stream.  Workspace must have been reserved (hjgpu_reserve).
 * Not valid inside a HIP stream capture (HJGPU_EINVAL on a capturing stream): a
 * replayed graph of a join faulted on gfx950 / ROCm 7.0. */
int  hjgpu_npj_async(hjgpu_ctx *ctx,
                     const uint32_t *d_inner_keys, const uint32_t *d_inner_vals, size_t inner,
                     const uint32_t *d_outer_keys, const uint32_t *d_outer_vals, size_t outer,
                     const hjgpu_npj_params *params, hjgpu_result *d_result, void *stream);
int  hjgpu_phj_async(hjgpu_ctx *ctx,
                     const uint32_t *d_inner_keys, const uint32_t *d_inner_vals, size_t inner,
                     const uint32_t *d_outer_keys, const uint32_t *d_outer_vals, size_t outer,
                     const hjgpu_phj_params *params, hjgpu_result *d_result, void *stream);
/* As hjgpu_phj_async, but the build columns only become valid once
 * `inner_ready_event` (a hipEvent_t passed as void*, recorded by the caller on the
 * stream that produces them, e.g. an RCCL broadcast) has fired: the probe side is
 * histogrammed and partitioned first, the wait sits right before the first kernel
 * that reads R.  This is how the multi-GPU PHJ hides the build-side broadcast.
 * (A grouped plan - see hjgpu_phj - reads R first: the stream waits for the event, and the CALLING thread waits for the groups.) */
int  hjgpu_phj_overlapped_async(hjgpu_ctx *ctx,
                     const uint32_t *d_inner_keys, const uint32_t *d_inner_vals, size_t inner,
                     const uint32_t *d_outer_keys, const uint32_t *d_outer_vals, size_t outer,
                     const hjgpu_phj_params *params, hjgpu_result *d_result, void *stream,
                     void *inner_ready_event);
int  hjgpu_cpra_async(hjgpu_ctx *ctx,
                      const uint32_t *d_inner_keys, const uint32_t *d_inner_vals, size_t inner,
                      const uint32_t *d_outer_keys, const uint32_t *d_outer_vals, size_t outer,
                      const hjgpu_phj_params *params, hjgpu_result *d_result, void *stream);
/* Status of an enqueue-only join.  The blocking forms report HJGPU_EZEROKEY (a build key of 0 met by NPJ: the
 * reference's empty-bucket sentinel, npj.cpp:196-210, 583, 867 - such a tuple is not in the table) and
 * HJGPU_EOVERFLOW (the reference's assert(o <= block_limit), npj.cpp:245); an *_async join cannot, so its caller asks:
 *   hjgpu_get_async_status        waits for `stream` and returns what the blocking form of the LAST join enqueued on this
 *                                 context would have returned (HJGPU_OK / HJGPU_EZEROKEY / HJGPU_EOVERFLOW);
 *   hjgpu_accumulate_async_status enqueue-only: d_flags[0] += 1 if that join met a zero build key, d_flags[1] += 1 if
 *                                 its materialised output overflowed (two uint64 in device memory: the multi-GPU joins
 *                                 all-reduce them with the four aggregates, so every rank returns the same status). */
int  hjgpu_get_async_status(hjgpu_ctx *ctx, void *stream);
int  hjgpu_accumulate_async_status(hjgpu_ctx *ctx, uint64_t *d_flags, void *stream);
/* Materialised rows from an enqueue-only join: the NEXT *_async join enqueued on this context (hjgpu_npj_async,
 * hjgpu_phj_async, hjgpu_phj_overlapped_async, hjgpu_cpra_async, hjgpu_phj_probe_async) writes its rows into `out`
 * exactly as the blocking form does (block protocol + close_gaps; the dense prefix [0, d_result->count) holds the
 * result once `stream` has passed the join).  One-shot; out == NULL withdraws it.  Overflow: see above. */
int  hjgpu_set_async_output(hjgpu_ctx *ctx, const hjgpu_output *out);
/* Capacity (in rows, a multiple of block_size) of result columns that are guaranteed to hold `rows` result rows
 * of a join of this context: rows rounded up to whole blocks plus one open block per worker (the reference sizes its
 * output the same way: 1.05 J + 2 T blocks, npj.cpp:997-1000).  algorithm: 0 npj (needs outer_tuples), 1 phj, 2 cpra;
 * block_size 0 = 65536. */
int  hjgpu_output_capacity(hjgpu_ctx *ctx, int algorithm, size_t outer_tuples, size_t rows, size_t block_size,
                           size_t *capacity);

/* ---- PHJ with the build side prepared once and probed by any number of batches -------------------
 * R join S = union over batches S_i of R join S_i, so a probe side that arrives in pieces (slices of a
 * multi-GPU exchange, batches from the host or from an upstream operator) needs the build side
 * histogrammed and partitioned only once: hjgpu_phj_build runs build-side K4/K5/K6 (the reference's
 * run_hj up to the end of the inner relation's passes, phj.cpp:1715-1863) and keeps the partitions in
 * the context's workspace; every hjgpu_phj_probe partitions one batch and runs build+probe per
 * partition (phj.cpp:1869-1924) against them.  `max_outer` sizes the workspace: batches may have up
 * to that many rows.  The prepared state lasts until any other join / partition entry point is called
 * on the context (hjgpu_phj_probe then fails with HJGPU_EINVAL); the caller's build columns are not
 * read again after hjgpu_phj_build has completed on `stream`.  Results are per batch: add them up. */
int  hjgpu_phj_build(hjgpu_ctx *ctx,
                     const uint32_t *d_inner_keys, const uint32_t *d_inner_vals, size_t inner,
                     size_t max_outer, const hjgpu_phj_params *params, void *stream);
int  hjgpu_phj_probe(hjgpu_ctx *ctx,
                     const uint32_t *d_outer_keys, const uint32_t *d_outer_vals, size_t outer,
                     hjgpu_result *result, const hjgpu_output *out, void *stream);
int  hjgpu_phj_probe_async(hjgpu_ctx *ctx,
                           const uint32_t *d_outer_keys, const uint32_t *d_outer_vals, size_t outer,
                           hjgpu_result *d_result, void *stream);

/* ---- PHJ over relations that ARRIVE pass-1-partitioned: the receiving side of the multi-GPU CPRA ---------------------
 * In the reference every worker partitions its own chunk and the owner of a partition gathers that partition's slice
 * from every chunk (cpra2.cpp:1757-1827 own-chunk passes, 1868-1872 ownership, 1891-1959 gather): the own-chunk
 * partitioning IS the join's partitioning.  Across GPUs the same holds when the exchange-level partitioning is pass 1:
 *   sender    hjgpu_partition_packed_async: own chunk -> packed tuples (payload << 32 | key, 8 bytes) partitioned with
 *             fan-out G * k, H(key, factor, G * k); rank g owns partitions [g * k, (g + 1) * k): ONE contiguous message
 *             per destination (keys and payloads travel together: one all-to-all-v instead of two);
 *   receiver  gets `chunks` pieces (one per source rank), each holding its k partitions in order, and runs pass 2 +
 *             build / probe only: hjgpu_phj_build_prepartitioned prepares the received build side once (fused histogram
 *             over the pieces, pass 2 into line-aligned final partitions), hjgpu_phj_probe_prepartitioned_async joins
 *             one received probe batch against it (results add up over the batches).
 * 16 bytes per tuple less HBM traffic on the receiver than partitioning the received tuples from scratch.
 * d_tuples_out of the sender must be 128-byte aligned; d_offsets: fanout + 1 uint64 (rows).  The receiver's pieces are
 * rows [chunk_offsets[c], chunk_offsets[c + 1]) of ONE array (contiguous, e.g. a receive buffer); a batch that is
 * too large for max_outer is passed in several calls, each a contiguous row range cut at any row (a piece of a
 * piece is still sorted by partition): pieces that do not take part have chunk_offsets[c] == chunk_offsets[c + 1]. */
typedef struct {
    uint32_t factor1;           /* the exchange-level pass: p1 = H(key, factor1, fanout1_total)                  */
    uint32_t fanout1_total;     /* G * k, <= 1024                                                                */
    uint32_t first_partition;   /* this rank owns pass-1 partitions [first_partition, first_partition + fanout1) */
    uint32_t fanout1;           /* k                                                                             */
    uint32_t chunks;            /* pieces (source ranks), 1..8                                                   */
    uint32_t reserved;
    uint64_t chunk_offsets[9];  /* rows; non-decreasing; entries beyond [chunks] are ignored                      */
} hjgpu_prepartitioned;
int  hjgpu_partition_packed_async(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                                  uint32_t factor, uint32_t fanout, uint64_t *d_tuples_out, uint64_t *d_offsets,
                                  void *stream);
/* The same, with the partitions [own_first, own_first + own_count) laid out LAST: rows [n - own_rows, n) of d_tuples_out,
 * behind all other partitions (which keep their order, rows [0, n - own_rows)).  d_offsets is still the plain prefix of
 * the counts (d_offsets[p + 1] - d_offsets[p] rows in partition p); a partition's first row follows from it:
 *   p <  own_first              d_offsets[p]
 *   p in the own range          n - own_rows + d_offsets[p] - d_offsets[own_first]
 *   p >= own_first + own_count  d_offsets[p] - own_rows              (own_rows = the own range's rows).
 * What it is for: a rank of the multi-GPU CPRA keeps its own partitions where they were written - the message to itself
 * (cpra2.cpp:1891-1959 gathers the owner's own chunk with the same memcpy as everybody else's) is never copied: the
 * other ranks' pieces are received right behind row n, and rows [n - own_rows, n + received) are the `chunks` pieces of
 * hjgpu_prepartitioned (the own piece first). */
int  hjgpu_partition_packed_own_last_async(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                                           uint32_t factor, uint32_t fanout, uint32_t own_first, uint32_t own_count,
                                           uint64_t *d_tuples_out, uint64_t *d_offsets, void *stream);
int  hjgpu_phj_build_prepartitioned(hjgpu_ctx *ctx, const uint64_t *d_tuples, const hjgpu_prepartitioned *layout,
                                    size_t max_outer, const hjgpu_phj_params *params /* fanout2, factor2, table factors, flags */,
                                    void *stream);
int  hjgpu_phj_probe_prepartitioned_async(hjgpu_ctx *ctx, const uint64_t *d_tuples, const hjgpu_prepartitioned *layout,
                                          hjgpu_result *d_result, void *stream);
/* Counts published with the partitions (cpra2.cpp:1783-1840: every worker's histogram of its chunk is what the owners'
 * gather is planned from).  The SENDER's histogram pass can count the receivers' second level in the same read of its keys:
 * hjgpu_partition_packed_counted_async = hjgpu_partition_packed(_own_last)_async (own_count = 0: plain order) that also
 * leaves d_counts2[p1 * fanout2 + p2] = rows of the chunk with H(key, factor, fanout) = p1 and H(key, factor2, fanout2) = p2
 * (fanout * fanout2 <= 32768: the fused histogram lives in LDS).  A receiver that is handed, for every piece c, the rows
 * [first_partition * fanout2, (first_partition + fanout1) * fanout2) of that piece's sender's d_counts2 - d_counts
 * [c * fanout1 * fanout2 + ...], in the order of the pieces - joins the batch without a histogram pass of its own over what
 * arrived (hjgpu_phj_probe_prepartitioned_counted_async; the batch must be whole pieces, as counted).  fanout2 / factor2
 * have to be the prepared build side's: hjgpu_prepartitioned_plan says what hjgpu_phj_build_prepartitioned plans for a
 * build side of `inner` rows (params->fanout2 forces it: all receivers of one exchange need the same). */
int  hjgpu_partition_packed_counted_async(hjgpu_ctx *ctx, const uint32_t *d_keys, const uint32_t *d_vals, size_t n,
                                          uint32_t factor, uint32_t fanout, uint32_t own_first, uint32_t own_count,
                                          uint32_t factor2, uint32_t fanout2, uint64_t *d_tuples_out, uint64_t *d_offsets,
                                          uint64_t *d_counts2, void *stream);
int  hjgpu_phj_probe_prepartitioned_counted_async(hjgpu_ctx *ctx, const uint64_t *d_tuples, const hjgpu_prepartitioned *layout,
                                                  const uint64_t *d_counts, hjgpu_result *d_result, void *stream);
int  hjgpu_prepartitioned_plan(hjgpu_ctx *ctx, size_t inner, uint32_t fanout1, const hjgpu_phj_params *params,
                               uint32_t *fanout2, uint32_t *factor2);
/* The planning rule of hjgpu_phj / hjgpu_cpra for relations of these sizes (the reference plans its passes from the partition
 * count, phj.cpp:1791-1808): *groups = the groups of the grouped plan, 0 = one or two passes.  hjgpu_cpra_multi asks it, with a
 * rank's share of the relations' total sizes, which road its ranks take. */
int  hjgpu_grouped_plan(hjgpu_ctx *ctx, size_t inner, size_t outer, const hjgpu_phj_params *params, uint32_t *groups);

/* ---- whole joins on HOST columns (what the npj/phj/cpra mains call after
 * their fread()s, npj.cpp:1013-1039): upload, join, return aggregates.
 * Batches (option "host_batch": -1 = the default: hjgpu_join_host_rows always, hjgpu_join_host when whole columns and
 * their workspace would not fit the device's free memory; 0 = never; n = always, n rows per batch; a probe side of at
 * least two batches of 64 Mi rows): the build side is uploaded and prepared once, the
 * probe side travels in batches through two device buffers - batch i is joined while batch i + 1 is on the bus (R join S
 * = union over the batches); the device never holds the probe side as a whole (it may be larger than the device's
 * memory), the call costs the upload plus the last batch's join.  stats->batches says how many there were; ms_total is
 * the device time of the whole join - the build side's work plus every batch's join, as after a call without batches -
 * and the phase times are the last batch's scaled to that sum (NPJ: ms_build = the table's build).  hjgpu_join_host_rows works the
 * same way: every batch's rows are made dense on the device and go home - into page-locked columns by a copy kernel, so
 * that the DMA engines carry the upload only - while the next batch is joined; a batch that outgrows its share of
 * rows->capacity (x 1.25: a skewed probe side) is joined once more alone, into device columns made for exactly its rows;
 * only a result beyond rows->capacity sends the call down the whole-column path (it starts over and reports the rows needed).  In batches algorithm 2 (CPRA) joins
 * every batch as ONE chunk of the probe side (cpra2.cpp:1757-1827 partitions every chunk on its own: a batch is one);
 * params->chunks applies to calls without batches.
 * Otherwise (small probe sides, host_batch = 0) the columns are uploaded whole on their own
 * stream, probe side first, build side behind it; PHJ / CPRA partition the probe side while the build side is still
 * arriving.  Page-locked columns (hjgpu_host_alloc) are DMA'd where they are, pageable ones staged
 * through two 32 MiB page-locked buffers the context keeps.
 * stats->ms_upload is the wall-clock time of the upload, ms_total the device time of the join. */
int  hjgpu_join_host(hjgpu_ctx *ctx, int algorithm /* 0 npj, 1 phj, 2 cpra */,
                     const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
                     const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
                     const hjgpu_phj_params *phj_params, const hjgpu_npj_params *npj_params,
                     hjgpu_result *result, hjgpu_stats *stats);

/* As hjgpu_join_host, but the join is materialised (the reference's join_keys / join_outer_vals /
 * join_inner_vals columns, npj.cpp:997-1000, which its mains allocate on the host) and the dense
 * result [0, result->count) is copied back into the caller's three host columns.
 * Page-locked result columns (hjgpu_host_alloc) are filled by DMA, pageable ones through two pinned
 * staging buffers.  Returns HJGPU_EOVERFLOW, with result->count set, when the join has more rows
 * than rows->capacity: call again with columns of at least that many rows.  Row order is
 * unspecified, as in the reference (block claiming, npj.cpp:244-246). */
typedef struct {
    uint32_t *keys, *outer_vals, *inner_vals;   /* host columns of `capacity` uint32 each */
    size_t    capacity;                         /* in rows                                */
} hjgpu_host_rows;
int  hjgpu_join_host_rows(hjgpu_ctx *ctx, int algorithm /* 0 npj, 1 phj, 2 cpra */,
                          const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
                          const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
                          const hjgpu_phj_params *phj_params, const hjgpu_npj_params *npj_params,
                          const hjgpu_host_rows *rows, hjgpu_result *result, hjgpu_stats *stats);
/* hjgpu_join_host_rows into result columns that SEVERAL calls share (one per GPU of a node, each on its own context and
 * host thread: what hjgpu_join_host_rows_multi does for PHJ / NPJ): the rows of this call are appended where an atomic
 * fetch-add on *cursor puts them - a batch's dense rows as soon as they exist, so the calls' rows interleave and
 * [0, *cursor) is dense when all calls have returned (every reference worker appends its blocks to the shared output
 * the same way: npj.cpp:244-246, 312-316).  result = THIS call's aggregates.  Rows that do not fit (rows->capacity, or a
 * skewed batch's device columns) are counted, not written: HJGPU_EOVERFLOW, result->count exact, nothing starts over. */
int  hjgpu_join_host_rows_shared(hjgpu_ctx *ctx, int algorithm,
                                 const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
                                 const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
                                 const hjgpu_phj_params *phj_params, const hjgpu_npj_params *npj_params,
                                 const hjgpu_host_rows *rows, uint64_t *cursor, hjgpu_result *result, hjgpu_stats *stats);

/* ---- multi-GPU joins ------------------------------------------------------------------------------
 * The reference's cross-worker exchange is part of run_hj itself: phj.cpp:1715-1770 (thread-level pass: both
 * relations are exchanged between the threads) and cpra2.cpp:1861-1971 (thread t owns partitions
 * [t*P/T, (t+1)*P/T), cpra2.cpp:1868-1872, and gathers them from every thread's chunk by memcpy).  Here a
 * RANK is one GPU's share of a join and the exchange is RCCL over xGMI, called from this library (C++):
 *   hjgpu_phj_multi / hjgpu_npj_multi   build side replicated from `root` (scatter of 1/G slices over the root's
 *       G-1 links + all-gather, or one ring broadcast), probe side sharded (thread_beg / thread_end ranges with
 *       T = G, npj.cpp:516-529), complete local join per rank - the replication overlaps the probe side's
 *       partitioning -, ncclAllReduce of the four aggregates.  Valid because R join S = union_g (R join S_g).
 *   hjgpu_cpra_multi   both relations chunked over the ranks (cpra2.cpp:1737-1742); every rank partitions its own
 *       chunk with fan-out G (cpra2.cpp:1757-1827), the counts are all-gathered (cpra2.cpp:1834-1840: counts
 *       published, barrier), rank g receives partition g of every chunk by grouped ncclSend / ncclRecv
 *       (all-to-all-v; the memcpy gather cpra2.cpp:1891-1959), local PHJ, ncclAllReduce.  The probe side travels
 *       in `slices` pieces: slice i+1 is partitioned and slice i-1 joined (build side prepared once,
 *       hjgpu_phj_build / hjgpu_phj_probe) while slice i is on the links.
 * A communicator holds one or more LOCAL ranks (one host thread drives them: grouped RCCL calls):
 *   hjgpu_comm_create_local   every rank in this process, ranks[i] on devices[i]: what ./phj and ./cpra use
 *       (all visible GPUs).  transport RCCL = ncclCommInitAll; transport LOOPBACK = exchanges are hipMemcpyAsync
 *       between the ranks' buffers, ranks may share a device: the complete orchestration (ownership, counts,
 *       slicing, reductions) then runs at any world size on ONE GPU with the real kernels (tests), or across
 *       the GPUs of a node by peer copies.
 *   hjgpu_comm_create_rank    one rank per process (torchrun, MPI): ncclCommInitRank with the id that rank 0
 *       got from hjgpu_comm_get_id and passed around (bench.py: over torch.distributed's gloo store).
 * Every process must make the same sequence of *_multi calls with the same `slices` / root / |R| (collectives).
 * The calls return when the result is on the host; it is the GLOBAL result on every rank. */
typedef struct hjgpu_comm hjgpu_comm;
#define HJGPU_TRANSPORT_RCCL      0
#define HJGPU_TRANSPORT_LOOPBACK  1
#define HJGPU_ERCCL               8   /* RCCL error (text in hjgpu_comm_last_error)                */
typedef struct { char bytes[128]; } hjgpu_comm_id;          /* ncclUniqueId */

int  hjgpu_comm_create_local(int nranks, const int *devices, int transport, hjgpu_comm **comm);
int  hjgpu_comm_get_id(hjgpu_comm_id *id);
int  hjgpu_comm_create_rank(int device, int nranks, int rank, const hjgpu_comm_id *id, hjgpu_comm **comm);
int  hjgpu_comm_destroy(hjgpu_comm *comm);
/* comm == NULL: why the last hjgpu_comm_create_* / hjgpu_comm_get_id of THIS thread failed (the communicator that could
 * not be made no longer exists; RCCL's own text, e.g. of ncclCommInitRank, is kept here) */
const char *hjgpu_comm_last_error(const hjgpu_comm *comm);
/* world size, local ranks of this process, global rank of local rank 0 */
int  hjgpu_comm_size(const hjgpu_comm *comm, int *nranks, int *nlocal, int *first_rank);
/* the join context of a local rank: its device memory (hjgpu_malloc), generator, options, per-phase stats */
hjgpu_ctx *hjgpu_comm_ctx(hjgpu_comm *comm, int local_rank);
/* option "ring_broadcast" (0 / 1): replicate the build side with one ncclBroadcast instead of scatter +
 * all-gather; "max_message_bytes" (n): split larger point-to-point messages into pieces; "reserve_cus" (n): CUs
 * that the ranks' partitioning kernels leave free for RCCL's kernels (default 16 with RCCL and > 1 rank, else 0: free, K6 is not CU-bound);
 * "timeout_ms" (n, 0 = none; HJGPU_COMM_TIMEOUT_MS presets it): deadline of every host-side wait of the multi-GPU
 * calls.  The reference's workers meet at pthread barriers (cpra2.cpp:1834-1840, phj.cpp:1715-1770) and a worker
 * that never arrives hangs the program; here the streams are polled, RCCL is asked for asynchronous errors
 * (ncclCommGetAsyncError), and at the deadline the communicator is aborted (ncclCommAbort): the call returns
 * HJGPU_ERCCL naming the rank and stream that did not finish, every later call on the communicator fails fast, and
 * the process can exit.  "stall_rank" (k) / "stall_ms" (n): fault injection for tests, loopback transport only - rank
 * k arrives n ms late at every collective.  "exchange_in_place" (0 / 1, default 1): a CPRA rank keeps its own partitions
 * where its partitioning wrote them and receives the others' pieces behind them (no copy of the message to itself);
 * "cpra_two_level" (0 / 1): round 2's CPRA plan (used automatically beyond 8 ranks); "cpra_grouped" (0 / 1 / 2, default 1; the SAME value on every
 * rank: it decides whether a collective is issued, ranks that differ would mismatch their collectives): the ranks
 * agree, from the relations' total sizes (one all-reduce of two words before the build side's exchange), whether a rank's share needs a
 * grouped plan (hjgpu_grouped_plan on local rank 0's join context: set "group_from" / "group_inner" / "group_always" alike on every
 * rank); if so the exchange has fan-out ranks, the probe side travels in one slice and every rank runs a whole local join whose plan
 * groups.  That road costs one pass more than hjgpu_phj's grouped plan (its exchange is no pass of the join): 1 takes it where it still
 * pays (from ~9 table fills per partition on, e.g. 2 G x 8 G per rank), 2 wherever hjgpu_grouped_plan groups, 0 never; "self_via_rccl" (tests);
 * "debug_forensics" (0 / 1, hjgpu_comm_get_forensics below). */
int  hjgpu_comm_set_option(hjgpu_comm *comm, const char *name, const char *value);
/* What the communicator really is: the transport's own view of the world (ncclCommCount / ncclCommUserRank /
 * ncclCommCuDevice of local rank 0, ncclGetVersion), so that a result line can prove that N ranks talked over RCCL. */
typedef struct {
    int  nranks, nlocal, first_rank;
    char transport[16];        /* "rccl" | "loopback"                                          */
    int  rccl_version;         /* ncclGetVersion, e.g. 22707; 0 without RCCL                   */
    int  rccl_nranks;          /* ncclCommCount of local rank 0's communicator; -1 without     */
    int  rccl_rank;            /* ncclCommUserRank                                             */
    int  rccl_device;          /* ncclCommCuDevice                                             */
    int  timeout_ms;           /* option "timeout_ms"                                          */
    int  aborted;              /* 1: a deadline expired or RCCL reported an asynchronous error */
} hjgpu_comm_info;
int  hjgpu_comm_get_info(hjgpu_comm *comm, hjgpu_comm_info *info);
/* Preflight ("measure link bandwidth first"): 1 MB through every collective the joins use
 * (all-gather, all-to-all-v with a different count for every pair, all-reduce), each verified word for word on the
 * host; then `link_bytes` from every rank to the peer k places on, for k = 1 .. G-1 (all ranks at once, every
 * pair's own xGMI link), and to all peers at once (the CPRA exchange's shape).  Local rank 0's view.
 * Returns HJGPU_ERCCL when a collective delivered wrong data (ok_* say which). */
typedef struct {
    uint32_t nranks, rank;
    uint32_t ok_all_gather, ok_all_to_all, ok_all_reduce;
    float    ms_all_gather, ms_all_to_all, ms_all_reduce;      /* 1 MB per rank, includes RCCL's lazy set-up */
    uint64_t link_bytes;
    float    link_GBs[64];     /* [peer]: link_bytes sent to `peer` while every other rank sends to ITS peer at the same
                                  distance (second of two rounds); 0 for the rank itself                      */
    float    all_to_all_GBs;   /* bytes this rank SENDS per second when every rank sends link_bytes to every peer at once */
} hjgpu_preflight;
int  hjgpu_comm_preflight(hjgpu_comm *comm, size_t link_bytes, hjgpu_preflight *report);
/* Communicator option "debug_forensics" (0 / 1; diagnostics): option "audit" (hjgpu_audit_read above) on both contexts of
 * every local rank; hjgpu_cpra_multi then keeps the records of the step's exchange-level partitioning calls and of its
 * joins.  words[] = for every local rank {global rank, np, nj} followed by np + nj records of 32 words (the rank's
 * partitioning calls in order: build side, probe slices; then its joins: build, probe batches).  *count = the words there
 * are; nothing is copied when capacity is smaller.  tools/stress_cpra.py --forensics prints them for every wrong step:
 * the stage whose sums differ from its input's is the kernel that lost the tuples. */
int  hjgpu_comm_get_forensics(hjgpu_comm *comm, uint64_t *words, size_t capacity, size_t *count);
/* hjgpu_audit_recheck on both contexts of every local rank: words[] = for every local rank and context {global rank, context
 * (0 exchange-level partitioning, 1 join), n} followed by n x 9 words; *count = the words there are. */
int  hjgpu_comm_recheck(hjgpu_comm *comm, uint64_t *words, size_t capacity, size_t *count);
/* "debug_forensics" = 2: hjgpu_cpra_multi reads every slice's records on the host as soon as the slice's partitioning / join has finished -
 * before the next use of its buffers is enqueued (the overlap of a slice's join with the next slice's partitioning stays) - and a stage
 * whose sums differ from its input's is looked at again on the spot (hjgpu_audit_recheck).  words[] = per event {global rank, context,
 * slice, n, the call's record (32 words)} followed by n x 9 words.  Empty after a step whose records all agreed. */
int  hjgpu_comm_get_frozen(hjgpu_comm *comm, uint64_t *words, size_t capacity, size_t *count);
/* every local rank's streams drained, then a collective over all ranks */
int  hjgpu_comm_barrier(hjgpu_comm *comm);

/* One rank's share of the relations: columns on that rank's device (16-byte aligned).
 * hjgpu_phj_multi / hjgpu_npj_multi: `inner` = |R| on EVERY rank, the build columns are read on `root` only (NULL
 * elsewhere); d_outer_* = the rank's probe shard.  hjgpu_cpra_multi: the rank's chunk of both relations. */
typedef struct {
    const uint32_t *d_inner_keys, *d_inner_vals;
    size_t          inner;
    const uint32_t *d_outer_keys, *d_outer_vals;
    size_t          outer;
} hjgpu_shard;

typedef struct {
    float    ms_wall;            /* host clock around the call                                              */
    float    ms_exchange;        /* local rank 0: device time of its exchanges (replication / all-to-all-v)  */
    float    ms_partition;       /* local rank 0: exchange-level partitioning (CPRA)                         */
    float    ms_exchange_wait;   /* local rank 0: stream time its joins spent waiting for an exchange        */
    uint32_t joins;              /* local rank 0: local join calls whose phase times are in `join`           */
    uint32_t self_copies;        /* local rank 0, CPRA: exchanges whose message to ITSELF had to be copied (0 when every
                                    exchange ran in place: the own partitions stay where the partitioning wrote them)  */
    uint64_t tuples_joined;      /* local rank 0: tuples those joins read                                    */
    uint64_t bytes_sent;         /* local rank 0: bytes sent to OTHER ranks                                  */
    float    ms_upload;          /* hjgpu_join_host_multi: wall clock of the call outside the join step itself (cutting the
                                    columns, enqueueing every rank's uploads); 0 for device-resident shards     */
    float    ms_overlap;         /* hjgpu_join_host_multi, local rank 0: how long BEFORE the last byte of its upload
                                    arrived its first join kernel was already on the device (> 0: the partitioning
                                    of the probe shard overlaps the upload, as in hjgpu_join_host)              */
    hjgpu_stats join;            /* local rank 0: phase times of the MEASURED joins, summed: PHJ / NPJ the one local
                                    join; CPRA the build and the last probe batch (`joins`, `tuples_joined` count
                                    exactly those) - reading every slice's times would hold the host thread back  */
} hjgpu_multi_stats;

/* shards: one entry per LOCAL rank, in rank order */
int  hjgpu_phj_multi(hjgpu_comm *comm, const hjgpu_shard *shards, int root, const hjgpu_phj_params *params,
                     hjgpu_result *result, hjgpu_multi_stats *stats);
int  hjgpu_npj_multi(hjgpu_comm *comm, const hjgpu_shard *shards, int root, const hjgpu_npj_params *params,
                     hjgpu_result *result, hjgpu_multi_stats *stats);
int  hjgpu_cpra_multi(hjgpu_comm *comm, const hjgpu_shard *shards, const hjgpu_phj_params *params,
                      int slices /* 0 = 4 (1 in a world of one) */, hjgpu_result *result, hjgpu_multi_stats *stats);
/* Materialised rows (the reference's join_keys / join_outer_vals / join_inner_vals, which every worker writes:
 * npj.cpp:882-915, cpra2.cpp:1965-1982) through the multi-GPU joins: every rank materialises ITS share of the result
 * into its own three device columns (block protocol + close_gaps; CPRA: slice after slice, each behind the rows of
 * the slices before it) and reports how many dense rows it holds; the concatenation of the ranks' rows is the result
 * (SURVEY.md 8e).  `rows`: one entry per LOCAL rank.  When some rank's columns are too small EVERY rank returns
 * HJGPU_EOVERFLOW (the flags are all-reduced with the aggregates), result->count is the global row count and
 * rows[l].rows what rank l needs: size the columns with hjgpu_output_capacity and call again. */
typedef struct {
    hjgpu_output out;            /* this rank's result columns (device memory of the rank's device)         */
    uint64_t     rows;           /* OUT: rows [0, rows) of them hold the rank's share of the result          */
} hjgpu_shard_rows;
int  hjgpu_phj_multi_rows(hjgpu_comm *comm, const hjgpu_shard *shards, hjgpu_shard_rows *rows, int root,
                          const hjgpu_phj_params *params, hjgpu_result *result, hjgpu_multi_stats *stats);
int  hjgpu_npj_multi_rows(hjgpu_comm *comm, const hjgpu_shard *shards, hjgpu_shard_rows *rows, int root,
                          const hjgpu_npj_params *params, hjgpu_result *result, hjgpu_multi_stats *stats);
int  hjgpu_cpra_multi_rows(hjgpu_comm *comm, const hjgpu_shard *shards, hjgpu_shard_rows *rows,
                           const hjgpu_phj_params *params, int slices, hjgpu_result *result, hjgpu_multi_stats *stats);
/* As hjgpu_join_host on all ranks of a LOCAL communicator (what ./npj ./phj ./cpra call when several GPUs are
 * visible): the host columns are cut into the ranks' shares (thread_beg / thread_end, T = ranks), uploaded, joined.
 * algorithm: 0 npj, 1 phj (build side uploaded to rank 0 and replicated from there), 2 cpra (both sides chunked). */
int  hjgpu_join_host_multi(hjgpu_comm *comm, int algorithm,
                           const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
                           const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
                           const hjgpu_phj_params *phj_params, const hjgpu_npj_params *npj_params,
                           hjgpu_result *result, hjgpu_multi_stats *stats);
/* hjgpu_join_host_rows on all ranks of a local communicator: every rank materialises its share, the shares are copied
 * back to back into the caller's three host columns (rank 0's rows first).  HJGPU_EOVERFLOW with result->count set
 * when the result has more rows than rows->capacity.  (A rank whose own device columns turn out too small is run a
 * second time with exactly the size it reported: not the caller's concern.) */
int  hjgpu_join_host_rows_multi(hjgpu_comm *comm, int algorithm,
                                const uint32_t *inner_keys, const uint32_t *inner_vals, size_t inner,
                                const uint32_t *outer_keys, const uint32_t *outer_vals, size_t outer,
                                const hjgpu_phj_params *phj_params, const hjgpu_npj_params *npj_params,
                                const hjgpu_host_rows *rows, hjgpu_result *result, hjgpu_multi_stats *stats);

/* ---- data generator (write.cpp / generate_data_for_join, cpra2.cpp:1578-1696):
 * statistical contract only — non-zero build keys, unique when outer_total >= inner_total
 * (write.cpp's semantics otherwise: min(inner, outer) distinct keys, the build side repeats
 * them — BASELINE configs[0], 1 M probe x 16 M build, has 16 copies per key), probe keys drawn
 * from them (every build key at least once when outer >= inner), payload = key*factor,
 * both sides in pseudo-random order; counter-based, so any shard
 * [outer_begin, outer_begin+outer_count) of the probe side can be produced
 * independently on its own GPU. */
int  hjgpu_generate(hjgpu_ctx *ctx, uint64_t seed, size_t inner, size_t outer_total,
                    size_t outer_begin, size_t outer_count,
                    uint32_t inner_factor, uint32_t outer_factor,
                    uint32_t *d_inner_keys, uint32_t *d_inner_vals,   /* may be NULL */
                    uint32_t *d_outer_keys, uint32_t *d_outer_vals,   /* may be NULL */
                    void *stream);
/* Same relations, but also the build side in ranges: rows [inner_begin, +inner_count)
 * of the inner_total-tuple build relation (CPRA across GPUs: every GPU owns a chunk of
 * BOTH relations, cpra2.cpp:1737-1742). */
int  hjgpu_generate_range(hjgpu_ctx *ctx, uint64_t seed, size_t inner_total, size_t outer_total,
                          size_t inner_begin, size_t inner_count, size_t outer_begin, size_t outer_count,
                          uint32_t inner_factor, uint32_t outer_factor,
                          uint32_t *d_inner_keys, uint32_t *d_inner_vals,
                          uint32_t *d_outer_keys, uint32_t *d_outer_vals, void *stream);
/* As hjgpu_generate_range, but the repeat picks of the probe side follow a Zipf law of exponent
 * `zipf` over the build keys (0 = uniform): the `zipf` argument of ./write (write.cpp:1685-1686,
 * whose own Zipf walk is unfinished).  Every build key still appears at least once
 * when outer_total >= inner_total, so with unique build keys the join aggregates remain
 * count = |S| and the column sums of S. */
int  hjgpu_generate_zipf(hjgpu_ctx *ctx, uint64_t seed, size_t inner_total, size_t outer_total,
                         size_t inner_begin, size_t inner_count, size_t outer_begin, size_t outer_count,
                         uint32_t inner_factor, uint32_t outer_factor, double zipf,
                         uint32_t *d_inner_keys, uint32_t *d_inner_vals,
                         uint32_t *d_outer_keys, uint32_t *d_outer_vals, void *stream);
/* As hjgpu_generate_zipf, with write.cpp's SELECTIVITY (write.cpp:1685-1689: with d = min(inner, outer) distinct
 * keys per side, join_d = d * selectivity of them are common: the build side draws from unique[0, d), the probe
 * side from unique[d - join_d, 2d - join_d)), so a full-size workload can hold probe tuples WITHOUT a match.
 * `expected` (may be NULL; needs outer_total >= inner_total, i.e. unique build keys): the aggregates that the join
 * of the generated probe range [outer_begin, +outer_count) with the WHOLE build side must return, accumulated
 * while the tuples are generated ("accumulate the expected aggregates during generation");
 * the ranges of several GPUs add up. */
int  hjgpu_generate_select(hjgpu_ctx *ctx, uint64_t seed, size_t inner_total, size_t outer_total,
                           size_t inner_begin, size_t inner_count, size_t outer_begin, size_t outer_count,
                           uint32_t inner_factor, uint32_t outer_factor, double zipf, double selectivity,
                           uint32_t *d_inner_keys, uint32_t *d_inner_vals,
                           uint32_t *d_outer_keys, uint32_t *d_outer_vals, hjgpu_result *expected, void *stream);
/* sum over a column of key, key*f_a, key*f_b (mod 2^32 per term, uint64 sums):
 * the analytic join aggregates of a selectivity-1 workload.
 * The kernel is a plain 16-byte-load streaming read; hjgpu_get_stats().ms_total after
 * the call is its duration (bench.py's empirical streaming-read ceiling). */
int  hjgpu_column_sums(hjgpu_ctx *ctx, const uint32_t *d_keys, size_t n,
                       uint32_t f_a, uint32_t f_b, uint64_t sums[3], void *stream);

/* Measurement helper: duration of a plain streaming read (16-byte loads, nothing computed)
 * of `bytes` (a multiple of 64 KiB is read) at d_ptr: the empirical HBM-read ceiling of this
 * device that bench.py reports next to the kernels' rates. */
int  hjgpu_stream_read_ms(hjgpu_ctx *ctx, const void *d_ptr, size_t bytes, float *ms, void *stream);
/* Measurement helper: duration of `reads` independent pseudo-random 64-byte line reads out of the `bytes` at d_ptr
 * (64-byte aligned), four lanes per line and four lines in flight per quad - the NPJ probe's access shape without the
 * join (npj.cpp:216-364 gathers one bucket per lane the same way).  The empirical ceiling bench.py prices NPJ against:
 * out of a table that does not fit the L2 the limit is memory-side requests per second (53-59 G/s for any request size
 * from 16 to 128 bytes, profiles/r03_request_size.txt), not bytes. */
int  hjgpu_random_line_read_ms(hjgpu_ctx *ctx, const void *d_ptr, size_t bytes, size_t reads, float *ms, void *stream);
/* Measurement helper: duration of `ops` independent pseudo-random 8-byte compare-and-swaps (expected 0) into the `bytes` at
 * d_ptr, which are zeroed first (outside the timed span) - the NPJ build's access shape without the join (npj.cpp:196-210:
 * one CAS of an empty bucket per build tuple).  in_flight = 1, 2, 4 or 8 CAS per lane; load_first: a plain load of the
 * bucket before the CAS, as the build skips taken buckets.  The empirical ceiling bench.py prices the NPJ build against. */
int  hjgpu_random_cas_ms(hjgpu_ctx *ctx, void *d_ptr, size_t bytes, size_t ops, int in_flight, int load_first, float *ms, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HJGPU_H */
