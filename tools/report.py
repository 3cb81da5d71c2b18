#!/usr/bin/env python3
"""One table of the numbers DESIGN.md quotes, measured in one go on the GPU box:
    python tools/report.py > gpurun_out/report.md
Every row is a `bench.py` run (its JSON line is parsed); the checksums are verified by bench.py itself
(`checksum_ok`).  Boxes differ by a few percent: compare rows of ONE report with each other."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [
    ("PHJ 64 M x 1 G (headline)", []),
    ("PHJ 64 M x 1 G, rows materialised", ["--materialize"]),
    ("CPRA 64 M x 1 G, 8 chunks", ["--algo", "cpra"]),
    ("NPJ 64 M x 1 G", ["--algo", "npj", "--steps", "5"]),
    ("PHJ Zipf(1.0) probe side", ["--zipf", "1.0"]),
    ("PHJ Zipf(1.5) probe side", ["--zipf", "1.5"]),
    ("PHJ Zipf(2.0) probe side", ["--zipf", "2.0"]),
    ("PHJ 1000 x 100 M (broadcast join)", ["--inner", "1000", "--outer", "100000000"]),
    ("PHJ 4000 x 1 G (broadcast join)", ["--inner", "4000", "--outer", "1000000000"]),
    ("PHJ 6900 x 1 G (broadcast join, 16 K-slot table)", ["--inner", "6900", "--outer", "1000000000"]),
    ("PHJ 100 K x 1 G (one pass)", ["--inner", "100000", "--outer", "1000000000"]),
    ("PHJ 1 M x 1 G (one pass)", ["--inner", "1000000", "--outer", "1000000000"]),
    ("PHJ 8 M x 1 G (two passes)", ["--inner", "8000000", "--outer", "1000000000"]),
    ("PHJ 128 M x 2.2 G (16 K-slot tables)", ["--inner", "128000000", "--outer", "2200000000", "--steps", "5"]),
]


def main():
    print("| Workload | ms / step | Gtuples/s | checksum | fan-out | hist | plan | scatter 1 | scatter 2 | join |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, extra in ROWS:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--cpu-outer", "0"] + extra
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not lines:
            print("| %s | FAILED (%d) | | | | | | | | |" % (name, p.returncode), flush=True)
            continue
        b = json.loads(lines[-1])
        ph = b["phase_ms"]
        mat = b.get("materialized")
        ms = mat["ms_total"] if mat else b["ms_per_step"]
        rate = b["config"]["outer_tuples_per_gpu"] / ms / 1e6
        ok = (mat["rows_checksum_ok"] if mat else b["checksum_ok"])
        print("| %s | %.2f | %.1f | %s | %s | %.2f | %.2f | %.2f | %.2f | %.2f |" % (
            name, ms, rate, ok, "x".join(str(f) for f in b["config"]["fanout"]), ph["ms_histogram"], ph["ms_plan"],
            ph["ms_scatter1"], ph["ms_scatter2"], (mat["ms_join"] if mat else ph["ms_join"]) + ph["ms_build"]), flush=True)


if __name__ == "__main__":
    main()
