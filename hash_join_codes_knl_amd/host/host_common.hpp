// host_common.hpp — shared by the npj / phj / cpra / write host programs.
//
// These programs keep the reference's command lines and file format
// (npj.cpp:929-1125, phj.cpp:1959-2231, cpra2.cpp:2017-2231, write.cpp:1677-1888):
//     ./npj|phj|cpra [#threads] [outer_tuples] [inner_tuples] [ratio]
//     ./write        [#threads] [outer_tuples] [inner_tuples] [selectivity] [zipf]
// reading / writing raw little-endian uint32 columns ./ik_<inner>.txt ./iv_<inner>.txt
// ./ok_<outer>.txt ./ov_<outer>.txt in the current directory.  Everything below
// the fread()s is replaced by one call into libhjgpu (include/hjgpu.h); no HIP
// header is needed here.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "hjgpu.h"

namespace hjhost {

struct Args {
    int threads;          // accepted for CLI compatibility; the GPU path does not use it
    size_t outer, inner;
    double extra;         // ratio (join programs) / selectivity (write)
    double zipf;
};

inline Args parse(int argc, char **argv, double extra_default)
{
    Args a;
    // defaults of the reference: hardware_threads(), 200 M, 200 M (npj.cpp:932-935)
    a.threads = argc > 1 ? atoi(argv[1]) : (int)std::thread::hardware_concurrency();
    if (a.threads < 1) a.threads = 1;
    a.outer = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)200 * 1000 * 1000;
    a.inner = argc > 3 ? (size_t)atoll(argv[3]) : (size_t)200 * 1000 * 1000;
    a.extra = argc > 4 ? atof(argv[4]) : extra_default;
    a.zipf = argc > 5 ? atof(argv[5]) : 0.0;
    return a;
}

inline std::string column_path(const char *prefix, size_t tuples)
{
    return std::string("./") + prefix + "_" + std::to_string(tuples) + ".txt";
}

// The reference does not check fopen/fread (a missing file segfaults); a host
// program reports and exits with status 2 instead.
inline bool read_column(const std::string &path, size_t tuples, std::vector<uint32_t> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s (generate it with ./write)\n", path.c_str()); return false; }
    out.resize(tuples);
    const size_t got = tuples ? fread(out.data(), sizeof(uint32_t), tuples, f) : 0;
    fclose(f);
    if (got != tuples) { fprintf(stderr, "%s: expected %zu tuples, read %zu\n", path.c_str(), tuples, got); return false; }
    return true;
}

inline bool write_column(const std::string &path, const std::vector<uint32_t> &col)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot create %s\n", path.c_str()); return false; }
    const size_t put = col.empty() ? 0 : fwrite(col.data(), sizeof(uint32_t), col.size(), f);
    fclose(f);
    return put == col.size();
}

struct Relations {
    std::vector<uint32_t> ik, iv, ok, ov;
};

// Columns of the join programs live in page-locked memory (hjgpu_host_alloc): the fread()s land
// where the GPU can DMA from at the PCIe rate (the reference's mamalloc'd columns, npj.cpp:982-1000).
struct PinnedRelations {
    hjgpu_ctx *ctx = nullptr;
    uint32_t *col[4] = {nullptr, nullptr, nullptr, nullptr};     // ik, iv, ok, ov
    size_t inner = 0, outer = 0;
    ~PinnedRelations() { for (uint32_t *c : col) if (c) hjgpu_host_free(ctx, c); }
};

inline bool read_into(const std::string &path, size_t tuples, uint32_t *out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s (generate it with ./write)\n", path.c_str()); return false; }
    const size_t got = tuples ? fread(out, sizeof(uint32_t), tuples, f) : 0;
    fclose(f);
    if (got != tuples) { fprintf(stderr, "%s: expected %zu tuples, read %zu\n", path.c_str(), tuples, got); return false; }
    return true;
}

inline bool load_relations(const Args &a, Relations &r)
{
    return read_column(column_path("ik", a.inner), a.inner, r.ik) &&
           read_column(column_path("iv", a.inner), a.inner, r.iv) &&
           read_column(column_path("ok", a.outer), a.outer, r.ok) &&
           read_column(column_path("ov", a.outer), a.outer, r.ov);
}

// HJGPU_ROWS: the result columns that arrived on the host must add up to the aggregates the device computed;
// HJGPU_ROWS=<prefix> (anything but "1") also writes them as raw uint32 files <prefix>jk_<J>.txt, jo_, ji_.
inline int report_rows(const char *rows_env, uint32_t *const rows_col[3], const hjgpu_result *res, double ms_download)
{
    uint64_t sums[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i)
        for (size_t j = 0; j < res->count; ++j) sums[i] += rows_col[i][j];
    const bool same = sums[0] == res->sum_keys && sums[1] == res->sum_outer_vals && sums[2] == res->sum_inner_vals;
    const double down = ms_download * 1e-3;
    fprintf(stderr, "result rows on the host: %llu x 12 bytes in %.4f s (%.1f GB/s), column sums %s\n",
            (unsigned long long)res->count, down, down > 0 ? 12.0 * res->count / down / 1e9 : 0.0, same ? "match" : "DIFFER");
    if (!same) return HJGPU_EHIP;
    if (strcmp(rows_env, "1") != 0) {
        const char *name[3] = {"jk", "jo", "ji"};
        for (int i = 0; i < 3; ++i) {
            const std::string path = std::string(rows_env) + name[i] + "_" + std::to_string(res->count) + ".txt";
            FILE *f = fopen(path.c_str(), "wb");
            const size_t put = f && res->count ? fwrite(rows_col[i], sizeof(uint32_t), res->count, f) : 0;
            if (f) fclose(f);
            if (!f || put != res->count) { fprintf(stderr, "cannot write %s\n", path.c_str()); return -2; }
        }
    }
    return HJGPU_OK;
}

// Loads the four column files, runs one join on the GPU and prints the extended report on stderr;
// the reference's own stdout line is printed by each main in its own format.
// Exit codes of the mains: 2 = input files, 1 = no GPU / join failed.
inline int run_join(int algorithm, const Args &a, hjgpu_result *res, hjgpu_stats *st, double *exchange_seconds = nullptr)
{
    if (exchange_seconds) *exchange_seconds = 0.0;
    // the files are checked before the device is touched: a missing input is reported as such
    const char *prefix[4] = {"ik", "iv", "ok", "ov"};
    const size_t tuples[4] = {a.inner, a.inner, a.outer, a.outer};
    for (int i = 0; i < 4; ++i) {
        FILE *f = fopen(column_path(prefix[i], tuples[i]).c_str(), "rb");
        if (!f) { fprintf(stderr, "cannot open %s (generate it with ./write)\n", column_path(prefix[i], tuples[i]).c_str()); return -2; }
        fclose(f);
    }
    // ---- several GPUs: every visible device takes a share (HJGPU_DEVICES="0,2,5" picks them) -------------
    // The reference's #threads workers become the GPUs of the node: PHJ / NPJ replicate the build side and
    // shard the probe side, CPRA chunks both sides and co-partitions them (include/hjgpu.h, multi-GPU joins).
    // HJGPU_TRANSPORT=loopback with HJGPU_RANKS=<n> runs n ranks on ONE device (tests of this host path).
    std::vector<int> devices;
    int transport = HJGPU_TRANSPORT_RCCL;
    {
        int visible = 0;
        (void)hjgpu_device_count(&visible);
        const char *tr = getenv("HJGPU_TRANSPORT"), *rk = getenv("HJGPU_RANKS"), *dv = getenv("HJGPU_DEVICES");
        if (tr && strcmp(tr, "loopback") == 0) transport = HJGPU_TRANSPORT_LOOPBACK;
        if (dv && *dv) {
            for (const char *p = dv; *p;) {
                char *end = nullptr;
                const long d = strtol(p, &end, 10);
                if (end == p) break;
                devices.push_back((int)d);
                p = *end == ',' ? end + 1 : end;
            }
        } else {
            const int n = (transport == HJGPU_TRANSPORT_LOOPBACK && rk) ? atoi(rk) : visible;
            for (int i = 0; i < n; ++i) devices.push_back(transport == HJGPU_TRANSPORT_LOOPBACK ? i % (visible > 0 ? visible : 1) : i);
        }
    }
    const char *rows_req = getenv("HJGPU_ROWS");
    const bool rows_wanted = rows_req && *rows_req && strcmp(rows_req, "0") != 0;
    if (devices.size() > 1) {
        // said once, on stderr: the reference's #threads is not what decides the workers here
        fprintf(stderr, "%zu GPUs visible: the join runs on all of them (%s); HJGPU_DEVICES=<id> keeps it on one\n", devices.size(),
                algorithm == 2 ? "both sides chunked over the GPUs, #threads is not used" : "build side replicated, probe side sharded");
        hjgpu_comm *comm = nullptr;
        setenv("NCCL_SOCKET_IFNAME", "lo", 0);     // all ranks live in this process: RCCL's bootstrap needs loopback only
        int rc = hjgpu_comm_create_local((int)devices.size(), devices.data(), transport, &comm);
        if (rc != HJGPU_OK) {
            fprintf(stderr, "hjgpu_comm_create_local(%zu ranks): %s (%s)\n", devices.size(), hjgpu_status_string(rc), hjgpu_comm_last_error(nullptr));
            return rc;
        }
        // a GPU that never arrives at an exchange ends the program with a message instead of hanging it (the reference's
        // pthread barriers would wait forever): 5 minutes unless HJGPU_COMM_TIMEOUT_MS says otherwise
        if (!getenv("HJGPU_COMM_TIMEOUT_MS")) (void)hjgpu_comm_set_option(comm, "timeout_ms", "300000");
        hjgpu_ctx *ctx0 = hjgpu_comm_ctx(comm, 0);
        {
            PinnedRelations r;
            r.ctx = ctx0; r.inner = a.inner; r.outer = a.outer;
            for (int i = 0; i < 4 && rc == HJGPU_OK; ++i) {
                rc = hjgpu_host_alloc(ctx0, (void **)&r.col[i], tuples[i] * sizeof(uint32_t));
                if (rc != HJGPU_OK) fprintf(stderr, "host allocation failed: %s\n", hjgpu_last_error(ctx0));
                else if (!read_into(column_path(prefix[i], tuples[i]), tuples[i], r.col[i])) rc = -2;
            }
            hjgpu_multi_stats ms;
            memset(&ms, 0, sizeof(ms));
            uint32_t *rows_col[3] = {nullptr, nullptr, nullptr};
            if (rc == HJGPU_OK && !rows_wanted) {
                rc = hjgpu_join_host_multi(comm, algorithm, r.col[0], r.col[1], a.inner, r.col[2], r.col[3], a.outer,
                                           nullptr, nullptr, res, &ms);
                if (rc != HJGPU_OK) fprintf(stderr, "join failed: %s (%s)\n", hjgpu_status_string(rc), hjgpu_comm_last_error(comm));
            } else if (rc == HJGPU_OK) {
                // HJGPU_ROWS: every GPU materialises its share, the shares land back to back in three host columns
                // (the reference's join_keys / join_outer_vals / join_inner_vals, npj.cpp:997-1000); sized for 1.05 x the
                // expected matches, a larger result reports its size and is run again with columns of that size
                size_t cap = (size_t)((double)(a.outer > a.inner ? a.outer : a.inner) * 1.05) + 1;
                for (int attempt = 0; attempt < 2; ++attempt) {
                    for (int i = 0; i < 3 && rc == HJGPU_OK; ++i)
                        rc = hjgpu_host_alloc(ctx0, (void **)&rows_col[i], cap * sizeof(uint32_t));
                    if (rc != HJGPU_OK) { fprintf(stderr, "host allocation failed: %s\n", hjgpu_last_error(ctx0)); break; }
                    hjgpu_host_rows rows = {rows_col[0], rows_col[1], rows_col[2], cap};
                    rc = hjgpu_join_host_rows_multi(comm, algorithm, r.col[0], r.col[1], a.inner, r.col[2], r.col[3], a.outer,
                                                    nullptr, nullptr, &rows, res, &ms);
                    if (rc != HJGPU_EOVERFLOW || attempt == 1) break;
                    for (int i = 0; i < 3; ++i) { hjgpu_host_free(ctx0, rows_col[i]); rows_col[i] = nullptr; }
                    cap = res->count;
                    rc = HJGPU_OK;
                }
                if (rc != HJGPU_OK) fprintf(stderr, "join failed: %s (%s)\n", hjgpu_status_string(rc), hjgpu_comm_last_error(comm));
                if (rc == HJGPU_OK) rc = report_rows(rows_req, rows_col, res, ms.join.ms_download);
            }
            for (uint32_t *c : rows_col) if (c) hjgpu_host_free(ctx0, c);
            if (rc == HJGPU_OK) {
                *st = ms.join;
                st->ms_total = ms.ms_wall;                 // the step as the host saw it: exchange + local joins
                if (exchange_seconds) *exchange_seconds = ms.ms_exchange * 1e-3;
                fprintf(stderr, "%zu ranks (%s): %s\n", devices.size(), transport == HJGPU_TRANSPORT_LOOPBACK ? "loopback" : "RCCL",
                        algorithm == 2 ? "both sides chunked, co-partitioned by all-to-all-v" : "build side replicated, probe side sharded");
                fprintf(stderr, "join_tuples=%llu sum_keys=%llu sum_outer_vals=%llu sum_inner_vals=%llu\n",
                        (unsigned long long)res->count, (unsigned long long)res->sum_keys,
                        (unsigned long long)res->sum_outer_vals, (unsigned long long)res->sum_inner_vals);
                fprintf(stderr, "step %.4f s: %.2f Gtuples/s probe-side; rank 0: exchange %.4f s (%.1f MB sent), partitioning %.4f s, "
                                "%u measured local joins %.4f s, waited %.4f s for exchanges; its first join kernel ran %.4f s before its "
                                "upload had ended\n",
                        ms.ms_wall * 1e-3, ms.ms_wall > 0 ? a.outer / (ms.ms_wall * 1e-3) / 1e9 : 0.0, ms.ms_exchange * 1e-3,
                        ms.bytes_sent / 1e6, ms.ms_partition * 1e-3, ms.joins, ms.join.ms_total * 1e-3, ms.ms_exchange_wait * 1e-3,
                        ms.ms_overlap * 1e-3);
            }
        }   // pinned columns are released before the communicator (they belong to its rank-0 context)
        hjgpu_comm_destroy(comm);
        return rc;
    }
    hjgpu_ctx *ctx = nullptr;
    int rc = hjgpu_create(devices.empty() ? -1 : devices[0], &ctx);
    if (rc != HJGPU_OK) { fprintf(stderr, "hjgpu_create: %s\n", hjgpu_status_string(rc)); return rc; }
    {
        PinnedRelations r;
        r.ctx = ctx; r.inner = a.inner; r.outer = a.outer;
        for (int i = 0; i < 4 && rc == HJGPU_OK; ++i) {
            rc = hjgpu_host_alloc(ctx, (void **)&r.col[i], tuples[i] * sizeof(uint32_t));
            if (rc != HJGPU_OK) fprintf(stderr, "host allocation failed: %s\n", hjgpu_last_error(ctx));
            else if (!read_into(column_path(prefix[i], tuples[i]), tuples[i], r.col[i])) rc = -2;
        }
        hjgpu_phj_params pp;
        memset(&pp, 0, sizeof(pp));
        if (algorithm == 2) {
            // #threads = independently partitioned chunks (cpra2.cpp:1757-1827, 2023; the reference's runs used 129 and more); the library takes up to 256
            pp.chunks = (uint32_t)(a.threads >= 1 && a.threads <= 256 ? a.threads : (a.threads > 256 ? 256 : 8));
            if ((int)pp.chunks != a.threads)
                fprintf(stderr, "cpra: %d chunks requested, %u used (the result does not depend on the chunk count)\n", a.threads, pp.chunks);
        }
        hjgpu_npj_params np;
        memset(&np, 0, sizeof(np));
        // HJGPU_ROWS=1: materialise the join into three host columns, as the reference's mains do
        // (join_keys / join_outer_vals / join_inner_vals, npj.cpp:997-1000); HJGPU_ROWS=<prefix> also
        // writes them as raw uint32 files <prefix>jk_<J>.txt, <prefix>jo_<J>.txt, <prefix>ji_<J>.txt.
        const char *rows_env = getenv("HJGPU_ROWS");
        const bool want_rows = rows_env && *rows_env && strcmp(rows_env, "0") != 0;
        uint32_t *rows_col[3] = {nullptr, nullptr, nullptr};
        if (rc == HJGPU_OK && !want_rows) {
            rc = hjgpu_join_host(ctx, algorithm, r.col[0], r.col[1], a.inner, r.col[2], r.col[3], a.outer,
                                 &pp, &np, res, st);
            if (rc != HJGPU_OK)
                fprintf(stderr, "join failed: %s (%s)\n", hjgpu_status_string(rc), hjgpu_last_error(ctx));
        } else if (rc == HJGPU_OK) {
            // the reference sizes its result for 1.05 x the expected matches (npj.cpp:997); a join with
            // more rows than that reports its size and is run again with columns of that size
            size_t cap = (size_t)((double)(a.outer > a.inner ? a.outer : a.inner) * 1.05) + 1;
            for (int attempt = 0; attempt < 2; ++attempt) {
                for (int i = 0; i < 3 && rc == HJGPU_OK; ++i)
                    rc = hjgpu_host_alloc(ctx, (void **)&rows_col[i], cap * sizeof(uint32_t));
                if (rc != HJGPU_OK) { fprintf(stderr, "host allocation failed: %s\n", hjgpu_last_error(ctx)); break; }
                hjgpu_host_rows rows = {rows_col[0], rows_col[1], rows_col[2], cap};
                rc = hjgpu_join_host_rows(ctx, algorithm, r.col[0], r.col[1], a.inner, r.col[2], r.col[3], a.outer,
                                          &pp, &np, &rows, res, st);
                if (rc != HJGPU_EOVERFLOW || attempt == 1) break;
                for (int i = 0; i < 3; ++i) { hjgpu_host_free(ctx, rows_col[i]); rows_col[i] = nullptr; }
                cap = res->count;
                rc = HJGPU_OK;
            }
            if (rc != HJGPU_OK)
                fprintf(stderr, "join failed: %s (%s)\n", hjgpu_status_string(rc), hjgpu_last_error(ctx));
            if (rc == HJGPU_OK) rc = report_rows(rows_env, rows_col, res, st->ms_download);
        }
        for (uint32_t *c : rows_col) if (c) hjgpu_host_free(ctx, c);
        hjgpu_device_info info;
        if (rc == HJGPU_OK && hjgpu_get_device_info(ctx, &info) == HJGPU_OK) {
            const double sec = st->ms_total * 1e-3, up = st->ms_upload * 1e-3;
            const double n = (double)a.inner + (double)a.outer;
            fprintf(stderr, "device: %s (%s, %d CUs)\n", info.name, info.arch, info.compute_units);
            fprintf(stderr, "join_tuples=%llu sum_keys=%llu sum_outer_vals=%llu sum_inner_vals=%llu\n",
                    (unsigned long long)res->count, (unsigned long long)res->sum_keys,
                    (unsigned long long)res->sum_outer_vals, (unsigned long long)res->sum_inner_vals);
            fprintf(stderr, "device time %.4f s: %.2f Gtuples/s probe-side, %.1f GB/s of input columns\n",
                    sec, sec > 0 ? a.outer / sec / 1e9 : 0.0, sec > 0 ? 8.0 * n / sec / 1e9 : 0.0);
            fprintf(stderr, "upload of the four columns %.4f s (%.1f GB/s over PCIe, pipelined with the join)\n",
                    up, up > 0 ? 8.0 * n / up / 1e9 : 0.0);
        }
    }   // pinned columns are released before the context
    hjgpu_destroy(ctx);
    return rc;
}

// ---- generator pieces for ./write (own restatement of write.cpp's intent) -------
// MT19937 exactly as seeded by the reference (rand32_init, npj.cpp:138-148).
struct Rand32 {
    uint32_t num[625];
    size_t index;
    explicit Rand32(uint32_t seed)
    {
        num[0] = seed;
        for (size_t i = 0; i != 623; ++i) num[i + 1] = 0x6c078965u * (num[i] ^ (num[i] >> 30));
        index = 624;
    }
    uint32_t next()
    {
        if (index == 624) {
            size_t i = 0;
            for (; i != 227; ++i) {
                const uint32_t y = (num[i] & 0x80000000u) + (num[i + 1] & 0x7fffffffu);
                num[i] = num[i + 397] ^ (y >> 1) ^ (0x9908b0dfu & (0u - (y & 1u)));
            }
            num[624] = num[0];
            for (; i != 624; ++i) {
                const uint32_t y = (num[i] & 0x80000000u) + (num[i + 1] & 0x7fffffffu);
                num[i] = num[i - 227] ^ (y >> 1) ^ (0x9908b0dfu & (0u - (y & 1u)));
            }
            index = 0;
        }
        uint32_t y = num[index++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

inline bool odd_prime(uint64_t x)
{
    for (uint64_t d = 3; d * d <= x; d += 2)
        if (x % d == 0) return false;
    return true;
}

}  // namespace hjhost
