"""One device-only compile of a kernel source per test session (hipcc cross-compiles without a GPU): the ISA text and the
compiler's kernel-resource remarks come from the same run, shared by test_kernel_resources.py and test_store_policy_isa.py."""
import functools
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hash_join_codes_knl_amd", "csrc")


@functools.lru_cache(maxsize=None)
def compile_device(source):
    """(assembly text, {demangled kernel name: resources}) of csrc/<source> for gfx950 at the library's optimisation level"""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, source + ".s")
        p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-DHJGPU_KERNEL_HASH=\"isa\"", "--cuda-device-only", "-S",
                            "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, source), "-o", out],
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        text = open(out).read()
    rows, cur = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = m.group(2)
            rows[cur] = {}
        elif cur:
            rows[cur][m.group(1).split(" [")[0]] = m.group(2)
    names = list(rows)
    plain = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines() if names else []
    res = {}
    for mangled, name in zip(names, plain):
        r = rows[mangled]
        res[name.strip()] = dict(vgpr=int(r.get("VGPRs", 0)), vspill=int(r.get("VGPRs Spill", 0)), sspill=int(r.get("SGPRs Spill", 0)),
                                 scratch=int(r.get("ScratchSize", 0)), occ=int(r.get("Occupancy", 0)))
    return text, res
