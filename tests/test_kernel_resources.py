"""CPU test (hipcc cross-compiles without a GPU): register allocation facts that the kernels' correctness or speed
rest on, read from the compiler's own remarks (tools/kernel_resources.py)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(source, flt):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"),
                          os.path.join(ROOT, "hash_join_codes_knl_amd", "csrc", source), flt],
                         capture_output=True, text=True, check=True).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"(.*?)\s+vgpr\s+(\d+)\s+spill v(\d+) s(\d+)\s+scratch (\d+)\s+occ (\d+)", line)
        if m:
            rows[m.group(1).strip()] = dict(vgpr=int(m.group(2)), vspill=int(m.group(3)), sspill=int(m.group(4)),
                                            scratch=int(m.group(5)), occ=int(m.group(6)))
    return rows


def test_hand_pipelined_join_instances_do_not_spill():
    """join_kernel<..., PIPE = true> keeps loads in flight behind the compiler's back (inline assembly, waits placed by
    hand): a VGPR spill could store an in-flight register before its data has arrived.  The instances that are
    launched must have no VGPR spills and no scratch, at the occupancy the geometry is planned for."""
    rows = resources("join_kernels.hip", "true, false, true")
    assert len(rows) == 2, rows
    for name, r in rows.items():
        assert r["vspill"] == 0 and r["scratch"] == 0 and r["occ"] == 4, (name, r)
