#!/bin/bash
# A longer validation sweep on the final library (beyond tools/r04_round.sh's): other seeds, 2 000 random joins (each also by a grouped
# plan and through the prepared-build and materialising paths), 800 random multi-GPU joins at loopback worlds, the slice pipeline's stress.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_validation_long.txt
hash=$(python3 -c "import hash_join_codes_knl_amd as H; from hash_join_codes_knl_amd import build; assert H.kernel_hash() == build.kernel_hash(); print(H.kernel_hash())") || exit 1
{ echo "# tools/r04_validation_long.sh, kernel hash $hash, $(date -u +%FT%RZ)"
  for seed in 31001 31002; do
    echo "HJ_FUZZ_SEED=$seed HJ_FUZZ_CASES=1000 tests/test_gpu_fuzz.py:"
    HJ_FUZZ_SEED=$seed HJ_FUZZ_CASES=1000 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -1
  done
  echo "HJ_FUZZ_SEED=31003 HJ_FUZZ_CASES=800 tests/test_gpu_multi.py -k random_multi:"
  HJ_FUZZ_SEED=31003 HJ_FUZZ_CASES=800 timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -m gpu -q -k random_multi 2>&1 | tail -1
  for o in "--steps 1500" "--steps 800 --transport loopback --world 3 --slices 5" "--steps 800 --option cpra_fused_counts=0"; do
    timeout -k 10 500 python tools/stress_cpra.py $o 2>&1 | grep "steps wrong\|WRONG"
  done; } > $out 2>&1
cat $out
