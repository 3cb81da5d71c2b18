#!/usr/bin/env python3
"""Round 6, review item 2: the wait-drop check of tests/test_pipeline_ordering.py names a wait by (stream, ordinal among that stream's
waits).  This runs the same check - list the waits that add an edge, drop each, expect a report - with ONE build of the harness,
30 times pinned to one CPU and 30 times beside busy loops on every CPU (round 5's global wait counter failed 1 run in 5).
CPU only.  usage: python tools/r06_ordering_determinism.py [runs=30]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
SCENARIOS = [("cpra-host", 2, 0), ("cpra-host", 2, 0, "--grouped"), ("cpra", 2, 3)]


def check(exe, scenario, prefix):
    def run(*extra):
        p = subprocess.run(prefix + [exe] + [str(a) for a in scenario] + list(extra), env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return p.returncode, p.stdout
    rc, out = run("--list-waits")
    if rc != 0:
        return "the clean run failed"
    fresh = []
    for w in re.search(r"^waits:(.*)$", out, re.M).group(1).split():
        k, edge, kind = w.split(":")
        dst, src = re.match(r"s(-?\d+)<-s(-?\d+)", edge).groups()
        if kind == "new" and (int(dst) - 1) // 4 == (int(src) - 1) // 4:
            fresh.append(k)
    missed = [k for k in fresh if run("--drop-wait", k)[0] == 0]
    return "dropped waits that went unnoticed: %r" % missed if missed else None


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "pipeline_ordering")
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-x", "c++", "-I", os.path.join(ROOT, "tests", "mock_hip"), "-fsanitize=address,undefined",
                               "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "cpp_pipeline_ordering.cpp"), "-o", exe, "-lpthread", "-ldl"])
        lines = ["# %d cpus; scenarios %r; every run lists the edge-adding waits and drops each of them" % (os.cpu_count(), SCENARIOS)]
        for label, prefix, busy in (("taskset -c 0", ["taskset", "-c", "0"], 0), ("beside %d busy loops" % os.cpu_count(), [], os.cpu_count())):
            hogs = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(busy)]
            try:
                failed = []
                for i in range(runs):
                    for sc in SCENARIOS:
                        why = check(exe, sc, prefix)
                        if why:
                            failed.append((i, sc, why))
            finally:
                for h in hogs:
                    h.kill()
            lines.append("%s: %d of %d runs passed%s" % (label, runs - len({i for i, _, _ in failed}), runs, "" if not failed else " - " + repr(failed[:3])))
            print(lines[-1], flush=True)
    open(os.path.join(ROOT, "profiles", "r06_ordering_determinism.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
