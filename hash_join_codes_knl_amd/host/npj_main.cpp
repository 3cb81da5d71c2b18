// ./npj [#threads] [outer_tuples] [inner_tuples] [ratio]  — npj.cpp:929-1125.
// stdout: "%.4f\n" seconds (npj.cpp:1114); stderr: "Phase N: pct (sec)" x3
// (npj.cpp:1104-1112) followed by the result aggregates.
#include "host_common.hpp"

int main(int argc, char **argv)
{
    const hjhost::Args a = hjhost::parse(argc, argv, 1.0);
    hjgpu_result res;
    hjgpu_stats st;
    const int rc = hjhost::run_join(0, a, &res, &st);     // loads the column files into pinned memory
    if (rc != HJGPU_OK) return rc == -2 ? 2 : 1;
    // Phase 1 = table init + build, Phase 2 = probe, Phase 3 = close_gaps (npj.cpp:878-915)
    const double ph[3] = {st.ms_build * 1e-3, st.ms_join * 1e-3, st.ms_close_gaps * 1e-3};
    const double total = st.ms_total * 1e-3;
    for (int p = 0; p < 3; ++p)
        fprintf(stderr, "Phase %ld: %5.2f%% (%.4f)\n", (long)p + 1, total > 0 ? ph[p] * 100.0 / total : 0.0, ph[p]);
    printf("%.4f\n", total);
    return 0;
}
