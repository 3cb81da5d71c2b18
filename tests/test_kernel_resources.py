"""CPU test (hipcc cross-compiles without a GPU): NO kernel instance of the library may have a private segment (scratch).

Round 3's multi-GPU stress runs (a rank's partitioning, exchange and join streams side by side) lost matches in ~5 % of
the steps while K6's pass-2 instance spilled 6 VGPRs.  Round 4 (DESIGN section 3 "Round 3 / 4",
profiles/r04_scratch_repro.txt) narrowed it down: a K6 launch with a private segment - even one word that is written once
and never read, the machine code otherwise identical - leaves ~10^-5 of its output slots unwritten when kernels of another
stream run beside it (tools/scratch_two_streams.py), never alone; the private values themselves are never wrong and the
cause is below the ISA.  So the rule is structural and for every kernel: zero scratch bytes per lane in every instance,
checked from the compiler's own remarks."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(source):
    """the compiler's kernel-resource remarks of one source (tests/device_compile.py: the same compile the ISA tests read;
    tools/kernel_resources.py prints the same table)"""
    from device_compile import compile_device
    return compile_device(source)[1]


@pytest.mark.parametrize("source", ["partition_kernels.hip", "join_kernels.hip", "npj_kernels.hip", "gen_kernels.hip",
                                    "audit_kernels.hip", "hjgpu_api.hip", "hjgpu_multi.hip"])
def test_no_kernel_instance_uses_scratch(source):
    rows = resources(source)
    assert rows, "no kernels found in %s" % source
    bad = {k: v for k, v in rows.items() if v["scratch"] or v["vspill"]}
    assert not bad, bad
