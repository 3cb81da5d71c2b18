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
    ("PHJ 64 M x 1 G (headline)", None),          # one run: also the materialised, CPRA and NPJ rows (bench.py's extras)
    ("PHJ Zipf(1.0) probe side", ["--zipf", "1.0"]),
    ("PHJ Zipf(1.5) probe side", ["--zipf", "1.5"]),
    ("PHJ Zipf(2.0) probe side", ["--zipf", "2.0"]),
    ("PHJ 1000 x 100 M (broadcast join)", ["--inner", "1000", "--outer", "100000000"]),
    ("PHJ 4000 x 1 G (broadcast join)", ["--inner", "4000", "--outer", "1000000000"]),
    ("PHJ 6900 x 1 G (broadcast join, 16 K-slot table)", ["--inner", "6900", "--outer", "1000000000"]),
    ("PHJ 100 K x 1 G (one pass)", ["--inner", "100000", "--outer", "1000000000"]),
    ("PHJ 1 M x 1 G (one pass)", ["--inner", "1000000", "--outer", "1000000000"]),
    ("PHJ 8 M x 1 G (two passes)", ["--inner", "8000000", "--outer", "1000000000"]),
    ("PHJ 200 M x 1 G (two table fills per partition)", ["--inner", "200000000", "--outer", "1000000000", "--steps", "5"]),
    ("PHJ 1 G x 4 G (grouped plan: pass 0 into 16 groups, then 16 two-pass joins)",
     ["--inner", "1000000000", "--outer", "4000000000", "--steps", "3", "--warmup", "1"]),
    ("PHJ 1 G x 4 G through hjgpu_phj_async (grouped plan enqueue-only: a worker thread of the context waits for the groups' sizes)",
     ["--inner", "1000000000", "--outer", "4000000000", "--steps", "3", "--warmup", "1", "--enqueue-only"]),
    ("PHJ 1 G x 4 G, option group_from=0 (two passes, 4-5 table fills per partition)",
     ["--inner", "1000000000", "--outer", "4000000000", "--steps", "3", "--warmup", "1", "--option", "group_from=0"]),
    ("PHJ 128 M x 2.2 G (16 K-slot tables)", ["--inner", "128000000", "--outer", "2200000000", "--steps", "5"]),
    ("CPRA 128 M x 2 G, 8 chunks (one GPU's share of configs[4] after the exchange)",
     ["--algo", "cpra", "--inner", "128000000", "--outer", "2000000000", "--steps", "5"]),
]


def row(name, ms, outer, ok, fan, ph, join_ms):
    print("| %s | %.2f | %.1f | %s | %s | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |" % (
        name, ms, outer / ms / 1e6, ok, fan, ph.get("ms_scatter0", 0), ph.get("ms_histogram", 0), ph.get("ms_plan", 0),
        ph.get("ms_scatter1", 0), ph.get("ms_scatter2", 0), join_ms), flush=True)


def main():
    print("| Workload | ms / step | Gtuples/s | checksum | fan-out | pass 0 | hist | plan | scatter 1 | scatter 2 | join |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, extra in ROWS:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--cpu-outer", "0"]
        cmd += ["--no-secondary"] + extra if extra is not None else []
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not lines:
            print("| %s | FAILED (%d) | | | | | | | | | |" % (name, p.returncode), flush=True)
            continue
        b = json.loads(lines[-1])
        ph = b["phase_ms"]
        outer = b["config"]["outer_tuples_per_gpu"]
        fan = "x".join(str(f) for f in b["config"]["fanout"])
        if b["config"].get("groups"):
            fan = "%d groups, each %s" % (b["config"]["groups"], fan)
        row(name, b["ms_per_step"], outer, b["checksum_ok"], fan, ph, ph["ms_join"] + ph["ms_build"])
        if extra is None:
            m, sec = b["materialized"], b["secondary"]
            row("PHJ 64 M x 1 G, rows materialised", m["ms_total"], outer, m["rows_checksum_ok"], fan, ph,
                m["ms_join"] + m["ms_close_gaps"])
            c = sec["cpra"]
            row("CPRA 64 M x 1 G, 8 chunks", c["ms_per_step"], outer, c["checksum_ok"], fan, c["phase_ms"], c["phase_ms"]["ms_join"])
            n = sec["npj"]
            row("NPJ 64 M x 1 G (build + probe)", n["ms_per_step"], outer, n["checksum_ok"], "-", {}, n["ms_build"] + n["ms_probe"])


if __name__ == "__main__":
    main()
