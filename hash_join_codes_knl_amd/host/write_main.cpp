// ./write [#threads] [outer_tuples] [inner_tuples] [selectivity] [zipf]
// Host data generator with the INTACT semantics of the reference's
// generate_data_for_join (cpra2.cpp:1578-1696; write.cpp:1482-1646 is a broken
// edit of it): unique non-zero keys drawn with MT19937 through a
// linear-probing set, every distinct key once then random repeats, global
// Fisher-Yates shuffle, payload = key * odd factor.  Deterministic: the seed is
// $HJ_SEED (default 1) instead of time(NULL) (write.cpp:1737); #threads is
// accepted and ignored (the reference generator is only reproducible at T = 1).
// zipf > 0 skews the repeat picks of the probe side (continuous inverse-CDF
// approximation of a Zipf law over the distinct keys).
// Files: ./ik_<inner>.txt ./iv_<inner>.txt ./ok_<outer>.txt ./ov_<outer>.txt
#include <math.h>

#include "host_common.hpp"

using hjhost::Rand32;

static void shuffle(std::vector<uint32_t> &d, Rand32 &gen)
{
    const size_t n = d.size();
    for (size_t i = 0; i != n; ++i) {
        uint64_t j = gen.next();
        j = ((j * (uint64_t)(n - i)) >> 32) + i;
        const uint32_t t = d[i]; d[i] = d[j]; d[j] = t;
    }
}

int main(int argc, char **argv)
{
    const hjhost::Args a = hjhost::parse(argc, argv, 1.0);
    const double selectivity = a.extra < 0 ? 0 : (a.extra > 1 ? 1 : a.extra);
    const char *se = getenv("HJ_SEED");
    const uint32_t seed = se ? (uint32_t)strtoul(se, nullptr, 10) : 1u;
    if (a.inner >= (1ull << 32) || a.outer >= (1ull << 32)) { fprintf(stderr, "relations must stay below 2^32 tuples\n"); return 2; }

    const size_t d = a.inner < a.outer ? a.inner : a.outer;          // write.cpp:1687-1689
    const size_t join_d = (size_t)((double)d * selectivity);
    const size_t distinct = 2 * d - join_d;
    Rand32 gen(seed);
    // factors: (rand() << 1) | 1 in the reference (cpra2.cpp:2069-2071); fixed streams here
    Rand32 fgen(seed ^ 0x5bd1e995u);
    const uint32_t unique_factor = fgen.next() | 1u, inner_factor = fgen.next() | 1u, outer_factor = fgen.next() | 1u;

    std::vector<uint32_t> uniq(distinct);
    {
        size_t buckets = distinct * 2 + 1;                           // cpra2.cpp:2087-2089
        while (!hjhost::odd_prime(buckets)) buckets += 2;
        std::vector<uint32_t> table(buckets, 0u);
        size_t i = 0;
        while (i != distinct) {                                       // unique(), cpra2.cpp:1544-1570
            uint32_t key;
            do key = gen.next(); while (key == 0);
            size_t h = (size_t)(((uint64_t)(uint32_t)(key * unique_factor) * buckets) >> 32);
            for (;;) {
                if (table[h] == key) break;
                if (table[h] == 0) { table[h] = key; uniq[i++] = key; break; }
                if (++h == buckets) h = 0;
            }
        }
    }
    hjhost::Relations r;
    r.ik.resize(a.inner); r.iv.resize(a.inner); r.ok.resize(a.outer); r.ov.resize(a.outer);
    size_t u = 0;
    for (size_t i = 0; i != a.inner; ++i) {                           // cpra2.cpp:1617-1624
        if (u != d) r.ik[i] = uniq[u++];
        else r.ik[i] = uniq[((uint64_t)gen.next() * d) >> 32];
    }
    const uint32_t *outer_unique = uniq.data() + (d - join_d);       // cpra2.cpp:1631
    u = 0;
    const double s = a.zipf;
    for (size_t o = 0; o != a.outer; ++o) {                           // cpra2.cpp:1639-1646
        if (u != d) { r.ok[o] = outer_unique[u++]; continue; }
        size_t pick;
        if (s <= 0) pick = (size_t)(((uint64_t)gen.next() * d) >> 32);
        else {
            const double x = (gen.next() + 0.5) / 4294967296.0;
            const double rank = (fabs(s - 1.0) < 1e-9) ? pow((double)d, x)
                                                       : pow((pow((double)d, 1.0 - s) - 1.0) * x + 1.0, 1.0 / (1.0 - s));
            pick = (size_t)rank - (rank >= 1.0 ? 1 : 0);
            if (pick >= d) pick = d - 1;
        }
        r.ok[o] = outer_unique[pick];
    }
    shuffle(r.ik, gen);                                               // cpra2.cpp:1654-1660
    shuffle(r.ok, gen);
    for (size_t i = 0; i != a.inner; ++i) r.iv[i] = r.ik[i] * inner_factor;     // cpra2.cpp:1663-1674
    for (size_t o = 0; o != a.outer; ++o) r.ov[o] = r.ok[o] * outer_factor;

    const bool ok = hjhost::write_column(hjhost::column_path("ik", a.inner), r.ik) &&
                    hjhost::write_column(hjhost::column_path("iv", a.inner), r.iv) &&
                    hjhost::write_column(hjhost::column_path("ok", a.outer), r.ok) &&
                    hjhost::write_column(hjhost::column_path("ov", a.outer), r.ov);
    fprintf(stderr, "wrote %zu inner / %zu outer tuples, %zu distinct keys, seed %u, factors %u %u\n",
            a.inner, a.outer, d, seed, inner_factor, outer_factor);
    return ok ? 0 : 2;
}
