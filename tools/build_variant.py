#!/usr/bin/env python3
"""Builds another libhjgpu.so next to the product's for A/B timing IN ONE PROCESS (tools/ab_libs.py): boxes and
even processes differ by up to 10 % in K6's pass 1, so only interleaved runs of two libraries in one process compare.
usage: python tools/build_variant.py <name> [<git-rev>] [-DMACRO=value ...]     (rev omitted: the working tree)
-> hash_join_codes_knl_amd/lib/variants/<name>.so"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hash_join_codes_knl_amd import build as B


def main():
    name = sys.argv[1]
    defines = [a for a in sys.argv[2:] if a.startswith("-D")]
    rest = [a for a in sys.argv[2:] if not a.startswith("-D")]
    rev = rest[0] if rest else None
    out_dir = os.path.join(B.LIB, "variants")
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "hash_join_codes_knl_amd", "csrc")
        os.makedirs(src)
        os.makedirs(os.path.join(tmp, "include"))
        files = ["include/hjgpu.h"] + ["hash_join_codes_knl_amd/csrc/" + f for f in os.listdir(B.CSRC)]
        for f in files:
            dst = os.path.join(tmp, f)
            if rev:
                try:
                    data = subprocess.check_output(["git", "show", "%s:%s" % (rev, f)], cwd=ROOT)
                except subprocess.CalledProcessError:
                    continue                      # the file does not exist in that revision
            else:
                data = open(os.path.join(ROOT, f), "rb").read()
            open(dst, "wb").write(data)
        objs = []
        for f in B.KERNEL_SOURCES:
            s = os.path.join(src, f)
            if not os.path.exists(s):
                continue
            o = os.path.join(tmp, f + ".o")
            subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-O3", "-std=c++20", "-fPIC", "-w",
                                   "-DHJGPU_KERNEL_HASH=\"%s\"" % name, "-DHJGPU_LIBRARY_HASH=\"%s\"" % name] + defines + ["-c", s, "-o", o])
            objs.append(o)
        so = os.path.join(out_dir, name + ".so")
        subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", so] + objs + ["-ldl"])
        print("built", so)


if __name__ == "__main__":
    main()
