"""The PRODUCT's multi-GPU orchestration (csrc/hjgpu_multi.hip, the same source file) on the CPU, under a recorder of stream
order, event edges and host-side waits (tests/cpp_pipeline_ordering.cpp + tests/mock_hip: a HIP runtime that records,
the single-GPU entry points as plain CPU code with the same contracts).  The reference's workers meet at barriers between
their phases (cpra2.cpp:1834-1840, phj.cpp:1715-1770) and cannot lose a tuple there; here a barrier is a
hipStreamWaitEvent or a host-side wait, and this test checks - without a GPU, at worlds 1 / 2 / 3 / 8 - that

* every read of a buffer is ordered after the buffer's last write, every write after its earlier readers and writers
  (stream order, an event edge, or a wait of the enqueuing host thread): 0 violations;
* the joins come out right (the mock really partitions, exchanges and joins: a tuple that reaches a rank that does not own
  its partition, a sender's histogram that differs from what arrived, a wrong aggregate or row are errors);
* the check has teeth: with ANY ONE hipStreamWaitEvent that adds an ordering edge removed (--drop-wait s<stream>#<n>), it reports.

Built with -fsanitize=address,undefined.  This replaces the test-side twin of round 1's orchestration as the CPU evidence
for SURVEY section 8 row (e) (tests/test_distributed_gloo.py keeps covering bench.py's gloo control plane)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


@pytest.fixture(scope="module")
def recorder(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pipeline_ordering") / "pipeline_ordering")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-x", "c++", "-I", os.path.join(ROOT, "tests", "mock_hip"),
                           "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           os.path.join(ROOT, "tests", "cpp_pipeline_ordering.cpp"), "-o", exe, "-lpthread", "-ldl"])
    return exe


def run(exe, *args):
    p = subprocess.run([exe] + [str(a) for a in args], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    return p.returncode, p.stdout


CPRA = [(w, s, o) for w in (1, 2, 3, 8) for s in (1, 3, 4, 8) for o in ((), ("--rows",))]
CPRA += [(w, s, o) for w in (1, 2, 3) for s in (1, 4)
         for o in (("--no-fused",), ("--no-in-place",), ("--no-fused", "--no-in-place"), ("--two-level",), ("--rows", "--no-fused"))]
# the grouped road (a rank's share beyond two passes' reach: sizes all-reduce, exchange in separate columns, one slice, whole local join)
CPRA += [(w, 4, o) for w in (1, 2, 3, 8) for o in (("--grouped",), ("--grouped", "--rows"))]


@pytest.mark.parametrize("world,slices,options", CPRA)
def test_cpra_exchange_and_slice_pipeline_are_ordered(recorder, world, slices, options):
    """hjgpu_cpra_multi(_rows): partition(i+1) | exchange(i) | join(i-1) on three streams per rank, two steps back to back"""
    rc, out = run(recorder, "cpra", world, slices, *options)
    assert rc == 0 and out.startswith("ok "), out
    assert "violations=0 errors=0 result=right" in out, out


@pytest.mark.parametrize("algo", ["phj", "npj"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("options", [(), ("--rows",)])
def test_replicated_build_side_joins_are_ordered(recorder, algo, world, options):
    """hjgpu_phj_multi / hjgpu_npj_multi(_rows): the build side replicated on the exchange stream, the probe shard partitioned meanwhile"""
    rc, out = run(recorder, algo, world, 1, *options)
    assert rc == 0 and "violations=0 errors=0 result=right" in out, out


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("options", [(), ("--rows",), ("--no-fused",), ("--grouped",)])
def test_cpra_from_host_columns_is_ordered(recorder, world, options):
    """hjgpu_join_host_multi / hjgpu_join_host_rows_multi, CPRA: the build side is uploaded first, the probe shard in the eight
    slices the join takes it in - slice i's partitioning waits for slice i's upload event only (cpra2.cpp:2110-2136 reads all
    four columns before anything starts)"""
    rc, out = run(recorder, "cpra-host", world, 0, *options)
    assert rc == 0 and "violations=0 errors=0 result=right" in out, out


@pytest.mark.parametrize("algo", ["phj-host", "npj-host"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_replicated_joins_from_host_columns_are_ordered(recorder, algo, world):
    """hjgpu_join_host_multi, PHJ / NPJ: every rank's shard on its own upload stream, probe side first; the probe shard is
    partitioned while the build columns are still arriving (the root's build columns gate only the replication)"""
    rc, out = run(recorder, algo, world, 0)
    assert rc == 0 and "violations=0 errors=0 result=right" in out, out


@pytest.mark.parametrize("scenario", [("cpra-host", 2, 0), ("phj-host", 3, 0), ("cpra", 1, 4), ("cpra", 2, 3), ("cpra", 3, 2, "--no-in-place"), ("cpra", 2, 2, "--rows"), ("cpra", 2, 4, "--grouped"), ("cpra-host", 2, 0, "--grouped"), ("phj", 3, 1), ("npj", 2, 1)])
def test_removing_any_wait_that_orders_something_is_reported(recorder, scenario):
    """Every hipStreamWaitEvent of the run that adds an edge (the waiting stream and the enqueuing host thread do not know the
    event's clock yet) is needed: without it the recorder reports an unordered access or the join comes out wrong.  The
    other waits are implied by a host-side wait made earlier (e.g. the host has waited for the partitioning stream before it
    enqueues the exchange) - they are listed, not required."""
    rc, out = run(recorder, *scenario, "--list-waits")
    assert rc == 0, out
    waits = re.search(r"^waits:(.*)$", out, re.M).group(1).split()
    # a rank's streams are made four at a time (join, partitioning, exchange, upload): stream ids 4 r + 1 ... 4 r + 4.  Waits between
    # streams of DIFFERENT ranks are the loopback transport's own fences around its copies (test infrastructure, conservative:
    # every rank waits for every other before and after); the orchestration's waits are between streams of one rank.
    def rank_of(stream):
        return (int(stream) - 1) // 4
    fresh = []
    for w in waits:
        k, edge, kind = w.split(":")
        dst, src = re.match(r"s(-?\d+)<-s(-?\d+)", edge).groups()
        if kind == "new" and rank_of(dst) == rank_of(src):
            fresh.append(k)                      # "s<stream>#<ordinal among that stream's waits>": the same wait in every run
    assert len(fresh) >= 4, waits
    undetected = []
    for k in fresh:
        rc, out = run(recorder, *scenario, "--drop-wait", k)
        if rc == 0:
            undetected.append(k)
    assert undetected == [], "dropped waits that went unnoticed: %r of %r" % (undetected, fresh)


def test_the_ranks_host_threads_are_race_free_under_tsan(tmp_path_factory):
    """The orchestration drives every local rank from a host thread of its own (each_rank): the same replay built with
    -fsanitize=thread - shared vectors of the step, the communicator's error text, the pinned scratch - must be silent."""
    exe = str(tmp_path_factory.mktemp("pipeline_ordering_tsan") / "pipeline_ordering_tsan")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-x", "c++", "-I", os.path.join(ROOT, "tests", "mock_hip"),
                           "-fsanitize=thread", os.path.join(ROOT, "tests", "cpp_pipeline_ordering.cpp"), "-o", exe, "-lpthread", "-ldl"])
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    for scenario in (("cpra", 2, 4), ("cpra", 3, 3, "--rows"), ("cpra", 8, 4), ("cpra", 2, 4, "--grouped"), ("cpra", 3, 4, "--no-fused", "--no-in-place"),
                     ("cpra", 2, 1, "--two-level"), ("phj", 3, 1), ("npj", 2, 1, "--rows"), ("cpra-host", 2, 0), ("cpra-host", 2, 0, "--grouped"),
                     ("phj-host", 3, 0), ("npj-host", 2, 0)):
        p = subprocess.run([exe] + [str(a) for a in scenario], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert p.returncode == 0 and "ThreadSanitizer" not in p.stdout, (scenario, p.stdout[-3000:])
        assert "violations=0 errors=0 result=right" in p.stdout, (scenario, p.stdout[-500:])
