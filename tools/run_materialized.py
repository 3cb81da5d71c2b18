#!/usr/bin/env python3
"""The materialising PHJ alone (3 result columns through the block protocol + close_gaps), for profiling:
    rocprofv3 --kernel-trace --stats ... -- python3 tools/run_materialized.py [steps] [solo]
Prints one JSON line with its times; bench.py reports the same leg as `materialized_default` (the library's default policy: rows through
non-temporal stores) and, with `solo`, as `materialized` (option solo: plain rows)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    torch.cuda.init()
    import hash_join_codes_knl_amd as H
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    inner, outer = 64_000_000, 1_000_000_000
    fi, fo = 0x2545F491, 0x9E3779B1
    with H.HjGpu(0) as hj:
        if "solo" in sys.argv[2:]:
            hj.set_option("solo", "1")   # bench.py's `materialized` leg: a blocking join in a process that runs nothing else (plain rows)
        ik, iv, ok, ov = hj.column(inner), hj.column(inner), hj.column(outer), hj.column(outer)
        hj.generate(1, inner, outer, 0, outer, fi, fo, ik, iv, ok, ov)
        sums = hj.column_sums(ok, outer, fo, fi)
        want = (outer, sums[0], sums[1], sums[2])
        block = 4096
        cap = ((outer + block - 1) // block + 8192 + 8) * block
        jk, jo, ji = (hj.column(cap, placed=True) for _ in range(3))      # as bench.py's `materialized` leg
        hj.reserve(inner, outer)
        t = {"ms_join": [], "ms_close_gaps": [], "ms_total": []}
        for i in range(steps + 1):
            got = hj.phj(ik, iv, inner, ok, ov, outer, out=(jk, jo, ji, cap, block))
            assert got == want
            if i:
                st = hj.stats()
                for k in t:
                    t[k].append(st[k])
        assert hj.column_sums(jk, outer, 1, 1)[0] == want[1]
        print(json.dumps({"workload": "materialising PHJ 64M x 1G, 12 B per row", "steps": steps,
                          **{k: round(sum(v) / len(v), 4) for k, v in t.items()}, "kernel_hash": H.kernel_hash()}))


if __name__ == "__main__":
    main()
