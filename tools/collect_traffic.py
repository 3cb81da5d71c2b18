#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 PMC counters, as MI355X_MICROARCH.md
prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), values in
KiB, and on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream, so
    bytes_read    = 2 * FETCH_SIZE * 1024
    bytes_written =     WRITE_SIZE * 1024
Run on the GPU box (rocprofv3 gets the program itself after `--`):
    python tools/collect_traffic.py [bench.py args...]
    python tools/collect_traffic.py --materialized        (the materialising PHJ of tools/run_materialized.py instead)
Writes gpurun_out/traffic.json; copy it to the file bench.py names (TRAFFIC_FILE, profiles/rNN_traffic.json).
The file records the kernel-source hash of the library it was measured with (hjgpu_kernel_hash); bench.py attaches
the counters to its `roofline.traffic` field only when that equals the running library's."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, extra):
    out = os.path.join(ROOT, "gpurun_out", "traffic_" + counter)
    shutil.rmtree(out, ignore_errors=True)          # a previous run's dispatches must not be averaged in
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    if extra[:1] == ["--materialized"]:
        program = [os.path.join(ROOT, "tools", "run_materialized.py"), "2"]
    else:
        program = [os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-outer", "0", "--no-secondary"] + extra
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable] + program
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            per[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}


def main():
    extra = sys.argv[1:]
    fetch = run_pass("FETCH_SIZE", extra)
    write = run_pass("WRITE_SIZE", extra)
    res = {}
    sizes = {"--inner": 64_000_000, "--outer": 1_000_000_000}
    for i, a in enumerate(extra):
        if a in sizes and i + 1 < len(extra):
            sizes[a] = int(extra[i + 1])
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        raw = f * 1024.0
        read = 2.0 * raw
        note = "wide coalesced streams: FETCH_SIZE counts half of them (gfx950), read = 2 * raw"
        # NPJ walks a table with 64-byte random line reads, which ARE counted at their size; only the column stream of
        # the kernel (8 bytes per tuple) is tallied at half.  Doubling the whole counter would double the line reads too
        # (round 2's file said 136 GB per probe launch for 72 GB real): real = raw + the missing half of the stream.
        stream = {"npj_probe": 8.0 * sizes["--outer"], "npj_build": 8.0 * sizes["--inner"]}
        for prefix, stream_bytes in stream.items():
            if k.startswith(prefix):
                read = raw + stream_bytes / 2.0
                note = "64-byte line reads counted at their size + a %d-byte column stream counted at half: read = raw + stream / 2" % stream_bytes
        res[k] = {"launches_seen": nf, "read_bytes_per_launch": read,
                  "written_bytes_per_launch": w * 1024.0,
                  "hbm_bytes_per_launch": read + w * 1024.0,
                  "fetch_size_bytes_raw": raw, "read_correction": note}
    sys.path.insert(0, ROOT)
    import hash_join_codes_knl_amd as H
    out = {"kernel_hash": H.kernel_hash(),
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over "
                     + ("`tools/run_materialized.py 2` (every join launch materialises its three result columns)"
                        if extra[:1] == ["--materialized"] else "`bench.py --steps 2 --warmup 1 --cpu-outer 0 --no-secondary`") +
                     ", averaged over all launches of a kernel; "
                     "read = 2*FETCH_SIZE KiB (gfx950 correction), written = WRITE_SIZE KiB",
           "bench_args": extra, "kernels": res}
    path = os.path.join(ROOT, "gpurun_out", "traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    for k, v in res.items():
        if any(s in k for s in ("scatter", "join", "hist2", "npj", "gaps")):
            print("%-40s read %.3f GB  written %.3f GB per launch (%d launches)"
                  % (k[:40], v["read_bytes_per_launch"] / 1e9, v["written_bytes_per_launch"] / 1e9, v["launches_seen"]))


if __name__ == "__main__":
    main()
