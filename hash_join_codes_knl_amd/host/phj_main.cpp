// ./phj [#threads] [outer_tuples] [inner_tuples] [ratio]  — phj.cpp:1959-2231.
// stdout: "%lf\t%lf\t%lf\t\n" = max / thread-0 / thread-128 seconds (phj.cpp:2197);
// a GPU (or, with several visible, all of them: host_common.hpp) runs the whole join, so the three columns carry
// the same time.
#include "host_common.hpp"

int main(int argc, char **argv)
{
    const hjhost::Args a = hjhost::parse(argc, argv, 1.0);
    hjgpu_result res;
    hjgpu_stats st;
    const int rc = hjhost::run_join(1, a, &res, &st);     // loads the column files into pinned memory
    if (rc != HJGPU_OK) return rc == -2 ? 2 : 1;
    fprintf(stderr, "fan-out %u x %u; histogram %.4f s, scatter %.4f + %.4f s, join %.4f s\n",
            st.fanout1, st.fanout2, st.ms_histogram * 1e-3, st.ms_scatter1 * 1e-3,
            st.ms_scatter2 * 1e-3, st.ms_join * 1e-3);
    const double t = st.ms_total * 1e-3;
    printf("%lf\t%lf\t%lf\t\n", t, t, t);
    return 0;
}
