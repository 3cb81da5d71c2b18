"""The multi-GPU host logic on the real data path: distributed.py + GpuOps (C-ABI) over RCCL at world size 1,
in a child process (its own process group; the parent keeps its context)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_replicated_build_and_copartitioned_join_over_rccl():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, os.path.join(HERE, "_dist_gpu_worker.py"), str(port)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ALL OK" in p.stdout, p.stdout + p.stderr
    assert p.stdout.count(" OK") >= 5
