"""Builds libhjgpu.so (HIP kernels + C-ABI) and the host CLI programs for gfx950.

hipcc cross-compiles without a GPU.  Artefacts are written in-tree under
hash_join_codes_knl_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB = os.path.join(PKG, "lib")
ARCH = "gfx950"

KERNEL_SOURCES = ["partition_kernels.hip", "join_kernels.hip", "npj_kernels.hip",
                  "gen_kernels.hip", "audit_kernels.hip", "hjgpu_api.hip", "hjgpu_ops.hip", "hjgpu_host.hip", "hjgpu_multi.hip"]
HOST_PROGRAMS = {"npj": "npj_main.cpp", "phj": "phj_main.cpp", "cpra": "cpra_main.cpp",
                 "write": "write_main.cpp"}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libhjgpu)")


# the kernels AND the plan that launches them (fan-out defaults, grid geometry, placement, reserve_cus live in hjgpu_api.hip)
KERNEL_HASH_FILES = ["partition_kernels.hip", "join_kernels.hip", "npj_kernels.hip", "hj_device.hpp",
                     "hj_emit.hpp", "hj_internal.hpp", "hjgpu_ctx.hpp", "hjgpu_api.hip", "hjgpu_ops.hip"]


def kernel_hash():
    """Identifies the kernels a measurement was taken with (profiles/*_traffic.json carry it; bench.py attaches
    PMC traffic to its roofline only when it equals the running library's, hjgpu_kernel_hash())."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_HASH_FILES:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def library_hash():
    """Identifies the LIBRARY a piece of evidence was taken with: every source that goes into libhjgpu.so - csrc/*.hip, csrc/*.hpp and
    include/hjgpu.h (hjgpu_library_hash()).  The kernel hash above gates PMC traffic (what the kernels move); stress / validation
    logs and the bench line carry this one, so a change of the orchestration (hjgpu_multi.hip, hjgpu_host.hip) changes their header."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    with open(os.path.join(ROOT, "include", "hjgpu.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def lib_path():
    # HJGPU_LIBRARY: load another build of the same ABI (A/B timing of kernel variants)
    return os.environ.get("HJGPU_LIBRARY") or os.path.join(LIB, "libhjgpu.so")


def build_library(force=False, verbose=True):
    os.makedirs(LIB, exist_ok=True)
    so = lib_path()
    srcs = [os.path.join(CSRC, f) for f in KERNEL_SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(ROOT, "include", "hjgpu.h"))
    if not force and not _newer(so, deps):
        return so
    objs = []
    for src in srcs:
        obj = os.path.join(LIB, os.path.basename(src) + ".o")
        if force or _newer(obj, deps):
            cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++20", "-fPIC",
                   "-Wall", "-Wno-unused-function", "-DHJGPU_KERNEL_HASH=\"%s\"" % kernel_hash(), "-DHJGPU_LIBRARY_HASH=\"%s\"" % library_hash(), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    # RCCL (multi-GPU joins, csrc/hjgpu_multi.hip) is NOT linked: the first communicator that asks for the RCCL
    # transport binds librccl.so.1 with dlopen (the copy already in the process - torch ships one - else the
    # system's), so single-GPU users, the C-ABI tests and hosts on a box without RCCL load the library without it
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", so] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return so


def build_host(force=False, verbose=True, sanitize=False):
    """The reference's four programs (./npj ./phj ./cpra ./write) as thin C++ hosts
    over the C-ABI.  Plain g++: they do not include HIP headers.
    sanitize=True: the same sources with -fsanitize=address,undefined into lib/asan/ (CPU-side checks of argv handling,
    file I/O and the generator, tests/test_sanitizers.py; the library they link stays the product's)."""
    os.makedirs(LIB, exist_ok=True)
    if sanitize:
        return _build_host_into(os.path.join(LIB, "asan"), force, verbose,
                                ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"])
    return _build_host_into(LIB, force, verbose, ["-O2"])


def _build_host_into(out_dir, force, verbose, flags):
    os.makedirs(out_dir, exist_ok=True)
    built = []
    common = [os.path.join(HOST, "host_common.hpp"), os.path.join(ROOT, "include", "hjgpu.h")]
    for name, src in HOST_PROGRAMS.items():
        srcp = os.path.join(HOST, src)
        if not os.path.exists(srcp):
            continue
        exe = os.path.join(out_dir, name)
        if force or _newer(exe, [srcp, lib_path()] + common):
            cmd = ["g++"] + flags + ["-std=c++20", "-Wall", "-I", os.path.join(ROOT, "include"),
                   srcp, "-o", exe, "-L", LIB, "-lhjgpu", "-Wl,-rpath," + ("$ORIGIN" if out_dir == LIB else "$ORIGIN/.."), "-lpthread"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        built.append(exe)
    return built


def build_all(force=False, verbose=True):
    so = build_library(force, verbose)
    build_host(force, verbose)
    return so


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built", lib_path())
