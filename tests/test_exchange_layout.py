"""CPU test of the multi-GPU CPRA's message layout (hash_join_codes_knl_amd/csrc/exchange_layout.hpp: where every rank's
message to every other rank starts, where it is received, which rows are the receiver's pieces - with the own
partitions written last and never copied, or through the copying path).  The header is plain host arithmetic shared
with csrc/hjgpu_multi.hip; tests/cpp_exchange_layout.cpp plays whole exchanges of 1 ... 8 ranks with even, ragged,
empty and one-destination chunks on the host (cpra2.cpp:1868-1872 ownership, 1891-1959 gather)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exchanges_of_one_to_eight_ranks_on_the_host(tmp_path):
    exe = tmp_path / "exchange_layout"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined",
                           "-I", os.path.join(ROOT, "hash_join_codes_knl_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp_exchange_layout.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok:")
