"""CPU tests: the C-ABI library builds, loads, and exports every symbol that
include/hjgpu.h declares (no compute calls - there is no GPU here)."""
import ctypes as C
import os
import re

import hash_join_codes_knl_amd as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hjgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hjgpu_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(H.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = H.load_library()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_status_strings_and_no_silent_fallback():
    lib = H.load_library()
    assert lib.hjgpu_status_string(0) == b"ok"
    assert b"sentinel" in lib.hjgpu_status_string(5) or b"reserved" in lib.hjgpu_status_string(5)
    h = C.c_void_p()
    st = lib.hjgpu_create(-1, C.byref(h))
    if st == H.api.OK:          # a GPU is visible (GPU box): the context must be real
        assert h.value
        lib.hjgpu_destroy(h)
    else:                       # no GPU: loud failure, never a CPU path
        assert st == H.api.ENODEVICE and not h.value


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors have the sizes and field offsets gcc gives include/hjgpu.h's structs."""
    import subprocess
    from hash_join_codes_knl_amd import api
    pairs = [("hjgpu_result", H.Result), ("hjgpu_phj_params", H.PhjParams), ("hjgpu_npj_params", H.NpjParams),
             ("hjgpu_output", H.Output), ("hjgpu_stats", H.Stats), ("hjgpu_host_rows", api.HostRows),
             ("hjgpu_device_info", api.DeviceInfo), ("hjgpu_shard", api.Shard), ("hjgpu_multi_stats", api.MultiStats),
             ("hjgpu_comm_id", api.CommId), ("hjgpu_comm_info", api.CommInfo), ("hjgpu_preflight", api.Preflight),
             ("hjgpu_shard_rows", api.ShardRows)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hjgpu.h"', 'int main(void){']
    for cname, cls in pairs:
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        lines.append('printf("\\n");')
    lines.append('return 0;}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    for (cname, cls), line in zip(pairs, out):
        got = [int(x) for x in line.split()[1:]]
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert got == want, (cname, got, want)
    assert C.sizeof(H.Stats) == 104 and C.sizeof(api.HostRows) == 32


def test_library_does_not_link_rccl():
    """RCCL is bound with dlopen by the first RCCL communicator: loading libhjgpu.so (single-GPU users, these tests,
    hosts on a box without RCCL) must not need it."""
    import subprocess
    from hash_join_codes_knl_amd import build as B
    needed = subprocess.check_output(["readelf", "-d", B.lib_path()], text=True)
    assert "librccl" not in needed, needed


def test_comm_create_errors_keep_their_text():
    """A communicator that cannot be made no longer exists: its text is kept per thread, hjgpu_comm_last_error(NULL)."""
    lib = H.load_library()
    h = C.c_void_p()
    st = lib.hjgpu_comm_create_local(0, None, H.api.TRANSPORT_LOOPBACK, C.byref(h))
    assert st == H.api.EINVAL and not h.value
    assert b"1 to 1024 ranks" in lib.hjgpu_comm_last_error(None)


def test_product_package_never_uses_the_oracle():
    """Nothing under the product package imports, links, includes or calls the oracle
    (comments may mention it); the stand-in operators of the gloo tests live in tests/oracle_ops.py."""
    pkg = os.path.join(ROOT, "hash_join_codes_knl_amd")
    banned = [r"^\s*(import|from)\s+oracle\b", r"libhjoracle", r"libhjref", r"#\s*include\s*[<\"].*hj_oracle",
              r"\bhjo_[a-z0-9_]+\s*\(", r"\bhjref_[a-z0-9_]+\s*\("]
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                for pat in banned:
                    assert not re.search(pat, src, flags=re.M), (f, pat)


def test_every_option_is_documented():
    """hjgpu_set_option's names (the list hjgpu_create reads from the environment, csrc/partition_kernels.hip) all appear
    in README.md's option table and in include/hjgpu.h's comment on hjgpu_set_option"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "hash_join_codes_knl_amd", "csrc", "partition_kernels.hip")).read()
    names = re.findall(r'"(\w+)"', re.search(r"static const char \*const names\[\] = \{(.*?)\};", src, re.S).group(1))
    assert len(names) >= 20
    readme = open(os.path.join(root, "README.md")).read()
    header = open(os.path.join(root, "include", "hjgpu.h")).read()
    assert [n for n in names if "`" + n not in readme] == []
    assert [n for n in names if '"' + n + '"' not in header] == []
