// ./cpra [#threads] [outer_tuples] [inner_tuples]  — cpra2.cpp:2017-2231.
// stdout: "copy:\t%lf\n" (gather time, cpra2.cpp:1984) then "%lf\n" seconds
// (cpra2.cpp:2208).  One GPU: #threads = number of independently partitioned chunks
// (1..8), the per-partition gather is done in place by the join kernel, so the
// copy time is 0 by construction.  Several GPUs visible: every GPU takes a chunk of
// both relations and the gather is the all-to-all-v over xGMI (hjgpu_cpra_multi);
// "copy" is then rank 0's exchange time.
#include "host_common.hpp"

int main(int argc, char **argv)
{
    const hjhost::Args a = hjhost::parse(argc, argv, 1.0);
    hjgpu_result res;
    hjgpu_stats st;
    double copy = 0.0;                                     // the gather: rank 0's all-to-all-v time with several GPUs
    const int rc = hjhost::run_join(2, a, &res, &st, &copy);     // loads the column files into pinned memory
    if (rc != HJGPU_OK) return rc == -2 ? 2 : 1;
    printf("copy:\t%lf\n", copy);
    printf("%lf\n", st.ms_total * 1e-3);
    return 0;
}
