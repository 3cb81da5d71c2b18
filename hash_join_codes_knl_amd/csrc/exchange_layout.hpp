// exchange_layout.hpp - where the messages of one CPRA exchange lie (hjgpu_multi.hip, CpraStep::exchange).
// Plain host arithmetic, no HIP: tests/test_exchange_layout.py compiles it with g++ and plays whole exchanges of
// 1 ... 8 ranks with ragged chunks on the CPU (every tuple has to arrive exactly once, in the piece of its source).
//
// A rank partitions its chunk of n rows with fan-out G * per into packed tuples; `prefix` is the plain prefix of the
// partition counts (per == 0 stands for round 2's two-level plan: fan-out G, per = 1).  Rank g owns partitions
// [g * per, (g + 1) * per): ONE contiguous message per destination (cpra2.cpp:1868-1872 ownership, 1891-1959 gather).
//   own_last  the own partitions were written LAST (hjgpu_partition_packed_own_last_async): rows [n - own, n)
//   in place  the other ranks' pieces are received right behind them, rows [n, n + received from others), in rank
//             order - the message to itself is not sent at all; the receiver's pieces are rows
//             [n - own, n + others) of the SEND buffer, the own piece first.  Needs room for n + others rows.
//   else      every piece (the own one included) goes to the receive buffer in rank order, rows from 0.
#pragma once
#include <stdint.h>
#include <vector>

namespace hj_exchange {
typedef unsigned long long u64;

// Does CPRA's GROUPED ROAD pay for a rank's share of `in` build and `out` probe rows (hjgpu_multi.hip cpra_join, comm option
// "cpra_grouped" = 1)?  The road - exchange with fan-out ranks, one probe slice, a whole local join whose plan groups - costs one
// pass more than hjgpu_phj's grouped plan: its exchange is a pass over both relations that is no pass of the join (the one-level
// plan's exchange IS the join's pass 1), and exchange and join do not overlap.  So: hjgpu_phj's rule (hjgpu_api.hip
// grouped_groups) with two passes' worth of overhead instead of one - every table fill beyond the first probes a partition's probe
// rows again, ~3.2 ms per 10^9 probe rows and fill; a pass is ~5 ms per 10^9 rows of both relations.  `reach` = the build rows two
// passes bring down to single-fill tables (max_parts / 2 partitions of 16 K-slot tables at 0.85).  Measured at RCCL world 1
// (profiles/r05_bench_force_dist_cpra_700M_4G*.json): 700 M x 4 G 70.0 ms on the one-level plan (3 fills per partition), 105.8 ms on
// the grouped road - the rule says one-level there; it says grouped from ~9 fills on (2 G x 8 G per rank).
inline bool grouped_road_pays(u64 in, u64 out, u64 max_parts)
{
    const double reach = (double)(max_parts / 2) * 16384.0 * 0.85;
    const double fills = (double)in / reach;
    return (fills - 1.0) * 3.2 * (double)out >= 1.1 * 2.0 * 5.0 * ((double)in + (double)out);
}

// where this rank's message to every destination starts in its send buffer, and how long it is
inline void send_layout(const u64 *prefix, size_t per, int G, int me, u64 n, bool own_last, u64 *soff, u64 *scnt)
{
    const size_t k = per ? per : 1;
    for (int p = 0; p < G; ++p) { soff[p] = prefix[(size_t)p * k]; scnt[p] = prefix[(size_t)(p + 1) * k] - prefix[(size_t)p * k]; }
    if (own_last) {
        // own partitions last: the others close up, the own ones end the chunk's rows
        const u64 own = scnt[me];
        for (int p = me + 1; p < G; ++p) soff[p] -= own;
        soff[me] = n - own;
    }
}

struct Receive {
    bool in_place;           // the pieces are rows of the SEND buffer (else: of the receive buffer)
    u64 rows;                // tuples this rank owns after the exchange (its own included)
    u64 need;                // in place: rows the send buffer has to hold (n + what the others send)
};

// matrix[src * G + dst] = rows src sends to dst.  Fills roff / rcnt (what the transport is told to receive where;
// in place: rcnt[me] = 0) and pieces[0 ... G]: piece c = rows [pieces[c], pieces[c + 1]) of the array named by
// Receive::in_place.  `capacity_rows`: rows the send buffer can hold (0 = never in place).
inline Receive receive_layout(const u64 *matrix, int G, int me, u64 n, bool own_last, u64 capacity_rows,
                              u64 *roff, u64 *rcnt, u64 *pieces)
{
    Receive r;
    u64 at = 0;
    for (int p = 0; p < G; ++p) { rcnt[p] = matrix[(size_t)p * (size_t)G + (size_t)me]; roff[p] = at; at += rcnt[p]; }
    r.rows = at;
    const u64 own = rcnt[me];
    r.need = n + (at - own);
    r.in_place = own_last && r.need <= capacity_rows;
    if (r.in_place) {
        u64 row = n;
        pieces[0] = n - own; pieces[1] = n;
        int piece = 1;
        for (int p = 0; p < G; ++p) {
            if (p == me) continue;
            roff[p] = row; row += rcnt[p];
            pieces[++piece] = row;
        }
        roff[me] = n - own; rcnt[me] = 0;                       // no message to itself
    } else {
        for (int p = 0; p < G; ++p) pieces[p] = roff[p];
        pieces[G] = at;
    }
    return r;
}
}  // namespace hj_exchange
