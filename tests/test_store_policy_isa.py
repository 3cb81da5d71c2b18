"""CPU test (hipcc cross-compiles without a GPU): the store policy of round 5, read from the machine code.

K6's plain stores were lost beside other streams' kernels (DESIGN section 3 "Round 5"); since then every whole-line store of K6
and every result row is non-temporal, and K6's partial-line stores are non-temporal in every instance but the solo one.  A
run-time flag around those stores looked right in the source and was WRONG in the binary - the compiler merged the two branches
into one plain store - so the policy is a template parameter / a build-time macro, and this test reads the ISA:
  scatter_kernel<..., NTP = true>   no plain 8- or 16-byte global store at all
  scatter_kernel<..., NTP = false>  16-byte stores non-temporal, 8-byte stores plain (option solo)
  join_kernel<..., NTROWS = true>   every 4-byte row store non-temporal (NTROWS = false: plain, solo joins); npj_probe_line_kernel: non-temporal"""
import collections
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hash_join_codes_knl_amd", "csrc")


def isa(source, tmp_path):
    from device_compile import compile_device
    text = compile_device(source)[0]
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @", text, re.M):
        body = text[m.end():text.find("s_endpgm", m.end())]
        stores = collections.Counter()
        for line in body.splitlines():
            line = line.strip()
            if line.startswith("global_store_dword"):
                stores[(line.split()[0], line.endswith(" nt"))] += 1
        kernels[m.group(1)] = stores
    return kernels


def test_k6_store_policy_in_the_machine_code(tmp_path):
    kernels = {k: v for k, v in isa("partition_kernels.hip", tmp_path).items() if k.startswith("_Z14scatter_kernel")}
    assert len(kernels) >= 40
    seen = {True: 0, False: 0}
    for name, stores in kernels.items():
        flags = re.findall(r"Lb([01])E", name)                 # RANGED, IN_PACKED, OUT_PACKED, CARRY, NTP
        out_packed, ntp = flags[2] == "1", flags[4] == "1"
        if not out_packed:                                     # separate output columns: 4-byte stores, whole lines per wave
            assert stores[("global_store_dword", False)] == 0 and stores[("global_store_dword", True)] > 0, (name, stores)
            continue
        seen[ntp] += 1
        assert stores[("global_store_dwordx4", False)] == 0 and stores[("global_store_dwordx4", True)] > 0, (name, stores)
        if ntp:
            assert stores[("global_store_dwordx2", False)] == 0 and stores[("global_store_dwordx2", True)] > 0, (name, stores)
        else:
            assert stores[("global_store_dwordx2", True)] == 0 and stores[("global_store_dwordx2", False)] > 0, (name, stores)
    assert seen[True] >= 16 and seen[False] >= 16


def test_result_rows_follow_the_policy_in_the_machine_code(tmp_path):
    """join_kernel<..., NTROWS>: every 4-byte row store non-temporal in the instances every pipeline uses, plain in the solo ones"""
    kernels = {k: v for k, v in isa("join_kernels.hip", tmp_path).items() if k.startswith("_Z11join_kernel")}
    seen = {True: 0, False: 0}
    for name, stores in kernels.items():
        nt_rows = re.findall(r"Lb([01])E", name)[3] == "1"         # PACKED, UNIQUE, DEDUP, NTROWS
        seen[nt_rows] += 1
        assert stores[("global_store_dword", not nt_rows)] == 0 and stores[("global_store_dword", nt_rows)] >= 3, (name, stores)
    assert seen[True] >= 10 and seen[False] >= 10


def test_npj_rows_are_non_temporal_in_the_machine_code(tmp_path):
    kernels = {k: v for k, v in isa("npj_kernels.hip", tmp_path).items() if k.startswith("_Z21npj_probe_line_kernel")}
    assert kernels
    for name, stores in kernels.items():
        if "ILb0E" in name:                                       # the instances that do not materialise have no row stores
            continue
        assert stores[("global_store_dword", False)] == 0 and stores[("global_store_dword", True)] >= 3, (name, stores)
