#!/bin/bash
# Round 5: the product with non-temporal K6 stores: the new tests, one bench line, then the slice pipeline where wrong steps were
# most frequent with plain stores (all streams at one priority: 5 wrong steps in 25 000).
# usage (GPU box): bash tools/r05_nt_product.sh [steps]  -> gpurun_out/r05_nt_product.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05_nt_product.txt
mkdir -p gpurun_out
echo "# tools/r05_nt_product.sh, $(date -u +%FT%RZ)" > $OUT
{ timeout -k 10 600 python3 -m pytest tests/test_gpu_shapes.py tests/test_gpu_parity.py -m gpu -q -x -k "more_chunks or 64_chunks or host_programs_end_to_end or cpra" 2>&1 | tail -3; } >> $OUT
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-outer 0 --no-secondary > gpurun_out/r05_nt_bench.json 2> gpurun_out/r05_nt_bench.err
python3 - >> $OUT <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_nt_bench.json"))
    print("bench: %.3f ms per step, phases %s, checksum %s, workspace %s" % (d["ms_per_step"], d["phase_ms"], d["checksum_ok"], d["workspace"]))
except Exception as ex:
    print("bench failed: %r" % (ex,))
PY
echo "## HJGPU_DEBUG_FLAT_PRIORITIES=1 stress_cpra.py --steps ${1:-45000} --slices 8 (product: nt stores)" >> $OUT
HJGPU_DEBUG_FLAT_PRIORITIES=1 timeout -k 10 800 python3 tools/stress_cpra.py --steps ${1:-45000} --slices 8 2>&1 | grep --line-buffered -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tee -a $OUT | grep --line-buffered "steps"
grep -v "^\.\.\." $OUT
