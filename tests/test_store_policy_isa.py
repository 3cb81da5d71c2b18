"""CPU test (hipcc cross-compiles without a GPU): the library's store policy, read from the machine code.

K6's plain stores were LOST IN MEMORY beside other streams' kernels - 1.3-1.5 in 10^4 steps of the multi-stream pipelines; round 6
read the damaged partitions back wrong through hipMemcpy and through a fresh kernel with the device quiet
(profiles/r06_lost_or_stale.txt) - and never with non-temporal stores.  The mechanism is below the ISA, so nothing exempts the other
kernels: EVERY global store of EVERY kernel of the library is non-temporal (csrc/hj_device.hpp: hj_store; the clears and
device-to-device copies are the library's own kernels, not the runtime's plain-storing hipMemsetAsync / hipMemcpyAsync) except
  scatter_kernel<..., NTP = false>   option "solo": K6's 8-byte partial-line stores plain (its 16-byte whole lines stay non-temporal)
  join_kernel<..., NTROWS = false>   option "solo": the result rows (4- and 16-byte stores) of a blocking join that runs alone on the device
  fill_probe_kernel                  the placement search's timing fill of candidate allocations (their content is never read)
  random_cas_kernel                  hjgpu_random_cas_ms, a ceiling measurement (what it leaves behind is never read).
A run-time flag around a store looked right in the source and was WRONG in the binary once - the compiler merged the two branches
into one plain store - so the policy is a template parameter / an unconditional helper, and this test reads the ISA."""
import collections
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hash_join_codes_knl_amd", "csrc")
SOURCES = ["partition_kernels.hip", "join_kernels.hip", "npj_kernels.hip", "gen_kernels.hip", "audit_kernels.hip",
           "hjgpu_api.hip", "hjgpu_ops.hip", "hjgpu_host.hip", "hjgpu_multi.hip"]
STORE = re.compile(r"^(global|flat|buffer)_store_\w+")


def isa(source):
    """{mangled kernel name: Counter{(store instruction, non-temporal?): n}} of csrc/<source>"""
    from device_compile import compile_device
    text = compile_device(source)[0]
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @", text, re.M):
        body = text[m.end():text.find("s_endpgm", m.end())]
        stores = collections.Counter()
        for line in body.splitlines():
            line = line.split(";")[0].strip()
            hit = STORE.match(line)
            if hit:
                stores[(hit.group(0), " nt" in line)] += 1
        kernels[m.group(1)] = stores
    return kernels


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def allowed_plain(name, plain_name):
    """the plain stores a kernel instance may have: None = none at all, else the set of store instructions"""
    if name.startswith("_Z14scatter_kernel"):
        flags = re.findall(r"Lb([01])E", name)                 # RANGED, IN_PACKED, OUT_PACKED, CARRY, NTP
        return {"global_store_dwordx2"} if flags[2] == "1" and flags[4] == "0" else None
    if name.startswith("_Z11join_kernel"):
        return {"global_store_dword", "global_store_dwordx4"} if re.findall(r"Lb([01])E", name)[3] == "0" else None      # PACKED, UNIQUE, DEDUP, NTROWS
    if "fill_probe_kernel" in plain_name or "random_cas_kernel" in plain_name:
        return {"global_store_dwordx4", "global_store_dwordx2"}
    return None


@pytest.mark.parametrize("source", SOURCES)
def test_no_plain_global_store_outside_the_solo_instances(source):
    kernels = isa(source)
    names = demangle(list(kernels))
    bad = {}
    for name, stores in kernels.items():
        ok = allowed_plain(name, names[name]) or set()
        plain = {k[0]: n for k, n in stores.items() if not k[1] and k[0] not in ok}
        if plain:
            bad[names[name]] = plain
    assert not bad, "plain global stores (use hj_store / hj_zero_async / hj_copy_async):\n" + "\n".join("%s: %s" % kv for kv in sorted(bad.items()))


def test_k6_store_policy_in_the_machine_code():
    kernels = {k: v for k, v in isa("partition_kernels.hip").items() if k.startswith("_Z14scatter_kernel")}
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


def test_result_rows_follow_the_policy_in_the_machine_code():
    """join_kernel<..., NTROWS>: every 4-byte row store non-temporal in the instances every pipeline uses, plain in the solo ones;
    the end cursors (8-byte) non-temporal in both"""
    kernels = {k: v for k, v in isa("join_kernels.hip").items() if k.startswith("_Z11join_kernel")}
    seen = {True: 0, False: 0}
    for name, stores in kernels.items():
        nt_rows = re.findall(r"Lb([01])E", name)[3] == "1"         # PACKED, UNIQUE, DEDUP, NTROWS
        seen[nt_rows] += 1
        assert stores[("global_store_dword", not nt_rows)] == 0 and stores[("global_store_dword", nt_rows)] >= 3, (name, stores)
        # (four rows of a lane at once, EmitterT::emit4: one 16-byte store per column, same policy)
        assert stores[("global_store_dwordx4", not nt_rows)] == 0 and stores[("global_store_dwordx4", nt_rows)] >= 3, (name, stores)
        assert stores[("global_store_dwordx2", False)] == 0, (name, stores)
    assert seen[True] >= 10 and seen[False] >= 10


def test_rows_of_npj_and_of_close_gaps_are_non_temporal_in_the_machine_code():
    kernels = isa("npj_kernels.hip")
    probes = {k: v for k, v in kernels.items() if k.startswith("_Z21npj_probe_line_kernel")}
    assert probes
    for name, stores in probes.items():
        if "ILb0E" in name:                                       # the instances that do not materialise have no row stores
            continue
        assert stores[("global_store_dword", False)] == 0 and stores[("global_store_dword", True)] >= 3, (name, stores)
    moves = [v for k, v in kernels.items() if "close_gaps_copy_kernel" in k]
    assert moves and all(v[("global_store_dword", True)] >= 3 and v[("global_store_dword", False)] == 0 for v in moves), moves


def test_clears_and_copies_are_the_librarys_own_kernels():
    """no hipMemsetAsync / device-to-device hipMemcpyAsync on a launch path: the runtime's fill and copy kernels store plainly"""
    for f in sorted(os.listdir(CSRC)):
        text = open(os.path.join(CSRC, f)).read()
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"hipMemsetAsync|hipMemcpyDeviceToDevice", text):
            line = text[:m.start()].count("\n") + 1
            ctx = text[max(0, m.start() - 200):m.end() + 80]
            # the two fallbacks inside hj_zero_async / hj_copy_async (ranges that are not made of 4-byte words: no caller has one)
            # and the diagnostics-only phase profile of K6 (option scatter_prof, synchronises)
            assert ("& 3" in ctx and f == "gen_kernels.hip") or "prof" in ctx, "%s:%d uses the runtime's %s" % (f, line, m.group(0))
