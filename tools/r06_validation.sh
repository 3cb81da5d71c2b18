#!/bin/bash
# Round 6: a longer randomised parity sweep on the round's library (other draws than the committed suite): random joins against numpy / the oracle
# through every single-GPU entry point, random shapes through the multi-GPU entry points at loopback worlds.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=gpurun_out/r06_validation.txt
mkdir -p gpurun_out
hash=$(python3 -c "
import sys, hash_join_codes_knl_amd as H
from hash_join_codes_knl_amd import build
assert H.library_hash() == build.library_hash(), 'stale library'
print(H.library_hash())") || exit 1
echo "# validation sweep, library hash $hash, $(date -u +%FT%RZ)" > $out
echo "## HJ_FUZZ_SEED=9602 HJ_FUZZ_CASES=${1:-1500} tests/test_gpu_fuzz.py" >> $out
# (progress goes to a file under gpurun_out/ as it is made: a run that writes nothing for seven minutes is taken to be hung)
HJ_FUZZ_SEED=9602 HJ_FUZZ_CASES=${1:-1500} PYTHONUNBUFFERED=1 timeout -k 10 1000 python3 -u -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/r06_validation_fuzz.log 2>&1; tail -1 gpurun_out/r06_validation_fuzz.log >> $out
echo "## HJ_FUZZ_SEED=9603 HJ_FUZZ_CASES=${2:-400} tests/test_gpu_multi.py -k random_multi" >> $out
HJ_FUZZ_SEED=9603 HJ_FUZZ_CASES=${2:-400} PYTHONUNBUFFERED=1 timeout -k 10 500 python3 -u -m pytest tests/test_gpu_multi.py -m gpu -q -k random_multi > gpurun_out/r06_validation_multi.log 2>&1; tail -1 gpurun_out/r06_validation_multi.log >> $out
cat $out
