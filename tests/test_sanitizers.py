"""The CPU side under sanitizers (sanitizers stay on the CPU builds: the GPU pool offers none).

* oracle/libhjoracle_asan.so (make -C oracle asan: -fsanitize=address,undefined) runs the whole golden-vector suite
  (tests/test_oracle_golden.py: the reference's scalar operators' outputs, bit for bit) in a python started with the
  sanitizer runtime preloaded;
* oracle/libhjoracle_tsan.so (make -C oracle tsan) runs the pthreads restatements of run() / run_hj() - barriers, the CAS
  build into the shared NPJ table (npj.cpp:196-210), block claims (npj.cpp:244-246), CPRA's gather - with 1, 3 and 8 threads;
* the four C++ hosts built with -fsanitize=address,undefined (lib/asan/): ./write byte for byte against the plain build,
  the join programs' argument and file handling up to the point where a GPU is needed."""
import os
import subprocess
import sys

import numpy as np
import pytest

import hash_join_codes_knl_amd as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


def _runtime(name):
    path = subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()
    if not os.path.isabs(path):
        pytest.skip("gcc has no " + name)
    return path


def test_golden_vectors_under_address_and_undefined_behaviour_sanitizers():
    subprocess.check_call(["make", "-C", ORACLE, "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=_runtime("libasan.so"), ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
               HJ_ORACLE_LIB=os.path.join(ORACLE, "libhjoracle_asan.so"))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-q", "-x", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]


def test_pthreads_joins_under_thread_sanitizer(tmp_path):
    subprocess.check_call(["make", "-C", ORACLE, "-s", "tsan"])
    script = tmp_path / "joins.py"
    script.write_text(
        "import sys\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import oracle as O\n"
        "ik, iv, ok, ov = O.generate(60000, 15000, seed=5)\n"
        "want = O.join_definition(ik, iv, ok, ov)\n"
        "for threads in (1, 3, 8):\n"
        "    assert O.npj(ik, iv, ok, ov, threads=threads) == want\n"
        "    assert O.npj(ik, iv, ok, ov, threads=threads, materialize=True)[0] == want\n"
        "    assert O.phj(ik, iv, ok, ov, threads=threads, hash_table_limit=300) == want\n"
        "    assert O.cpra(ik, iv, ok, ov, threads=threads, num_partitions=64) == want\n"
        "print('joined', want)\n" % ROOT)
    env = dict(os.environ, LD_PRELOAD=_runtime("libtsan.so"), TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66",
               HJ_ORACLE_LIB=os.path.join(ORACLE, "libhjoracle_tsan.so"))
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "joined" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-4000:]


def test_host_programs_under_address_sanitizer(tmp_path):
    H.build.build_all(verbose=False)
    plain = os.path.join(os.path.dirname(os.path.abspath(H.__file__)), "lib")
    san = {os.path.basename(p): p for p in H.build.build_host(verbose=False, sanitize=True)}
    env = dict(os.environ, HJ_SEED="21", ASAN_OPTIONS="detect_leaks=1")
    for d, exe in (("plain", os.path.join(plain, "write")), ("asan", san["write"])):
        (tmp_path / d).mkdir()
        p = subprocess.run([exe, "4", "30000", "7000", "0.5", "0.8"], cwd=tmp_path / d, env=env, capture_output=True, text=True)
        assert p.returncode == 0 and "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    for name in ("ik_7000.txt", "iv_7000.txt", "ok_30000.txt", "ov_30000.txt"):
        a, b = np.fromfile(tmp_path / "plain" / name, dtype="<u4"), np.fromfile(tmp_path / "asan" / name, dtype="<u4")
        assert len(a) and np.array_equal(a, b), name
    # the join programs: usage errors, missing files, and (without a GPU) the loud failure - never a sanitizer report
    for prog in ("npj", "phj", "cpra"):
        for args, cwd in ((["8"], tmp_path), (["8", "30000", "7000"], tmp_path), (["64", "30000", "7000"], tmp_path / "asan")):
            p = subprocess.run([san[prog]] + args, cwd=cwd, env=env, capture_output=True, text=True)
            assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, (prog, args, p.stderr[-3000:])
            if len(args) < 3 or cwd == tmp_path:
                assert p.returncode != 0, (prog, args)
