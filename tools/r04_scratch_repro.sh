#!/bin/bash
# Round 4, review item 1: is a wave's scratch unsafe next to other queues' kernels, or does the slice pipeline race?
# (Round 5 removed the -DHJ_SCRATCH_EXPERIMENT variants from the kernel sources: build them from round 4's tree, which
# tools/build_variant.py does when it is given a revision - e.g. `python tools/build_variant.py scratch_exp1 49bf42e -DHJ_SCRATCH_EXPERIMENT=1`.)
# Run on the GPU box after building the variants HERE (CPU cross-compile):
#   for e in 1 2 3 4; do python tools/build_variant.py scratch_exp$e 49bf42e -DHJ_SCRATCH_EXPERIMENT=$e; done
#   hipcc --offload-arch=gfx950 -O3 tools/ubench_scratch_race.hip -o hash_join_codes_knl_amd/lib/ubench_scratch_race
# Everything (failing logs included) goes to gpurun_out/r04_scratch_repro.txt; nothing is filtered except RCCL's banner.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r04_scratch_repro.txt
STEPS=${STEPS:-150}
V=hash_join_codes_knl_amd/lib/variants
mkdir -p gpurun_out
{
echo "# r04 scratch reproduction, $(date -u +%Y-%m-%dT%H:%MZ)"
echo "## versions"
echo "ROCm $(cat /opt/rocm/.info/version 2>/dev/null), kernel $(uname -r), amdgpu module $(cat /sys/module/amdgpu/version 2>/dev/null || echo '(in-tree)')"
/opt/rocm/bin/rocminfo 2>/dev/null | grep -m3 -E 'Runtime Version|Marketing Name.*MI|Name: +gfx'
python3 -c "import torch; print('torch', torch.__version__, 'hip', torch.version.hip)" 2>/dev/null
echo
echo "## (a) stand-alone: kernels with scratch, no join code (tools/ubench_scratch_race.hip)"
timeout -k 10 120 hash_join_codes_knl_amd/lib/ubench_scratch_race ${UB_SECONDS:-6} 2>&1
echo "ubench rc=$?"
} > $OUT 2>&1

run() {   # run <title> <library or ""> <env assignments...> -- <stress args...>
  local title="$1" lib="$2"; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  {
    echo
    echo "### $title"
    echo "\$ ${envs[*]} ${lib:+HJGPU_LIBRARY=$lib} python tools/stress_cpra.py --steps $STEPS $*"
    env "${envs[@]}" ${lib:+HJGPU_LIBRARY=$PWD/$lib} timeout -k 10 300 python tools/stress_cpra.py --steps $STEPS "$@" 2>&1 \
      | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path"
    echo "rc=${PIPESTATUS[0]}"
  } >> $OUT 2>&1
}

echo >> $OUT; echo "## (b) the slice pipeline (hjgpu_cpra_multi, 64 M x 1 G, 8 slices), $STEPS checked steps per line" >> $OUT
run "product library, RCCL world 1" "" --
run "product library, loopback world 2" "" -- --transport loopback --world 2
run "variant 1 (pass 2 spills as before d8b72d0: 8 VGPRs, 36 B of scratch per lane), RCCL world 1" $V/scratch_exp1.so --
run "variant 1, exchange_in_place=0 (the pipeline of the round-3 bisect: self-message copied)" $V/scratch_exp1.so -- --option exchange_in_place=0
run "variant 1, loopback world 2" $V/scratch_exp1.so -- --transport loopback --world 2
run "variant 1, HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" $V/scratch_exp1.so HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 --
run "variant 1, HSA_NO_SCRATCH_RECLAIM=1" $V/scratch_exp1.so HSA_NO_SCRATCH_RECLAIM=1 --
run "variant 1, HSA_ENABLE_SCRATCH_ALT=0" $V/scratch_exp1.so HSA_ENABLE_SCRATCH_ALT=0 --
run "variant 1, GPU_MAX_HW_QUEUES=1" $V/scratch_exp1.so GPU_MAX_HW_QUEUES=1 --
run "variant 1, GPU_MAX_HW_QUEUES=8" $V/scratch_exp1.so GPU_MAX_HW_QUEUES=8 --
run "variant 1, debug_serialize=1 (host waits after every slice's join)" $V/scratch_exp1.so -- --option debug_serialize=1
run "variant 1, debug_serialize=2 (partitioning waits for the joins enqueued so far)" $V/scratch_exp1.so -- --option debug_serialize=2
run "variant 1, debug_serialize=4 (exchange waits for the joins enqueued so far)" $V/scratch_exp1.so -- --option debug_serialize=4
echo >> $OUT; echo "## (c) scratch by construction, checked inside the kernel (volatile private copies of addresses, used and compared)" >> $OUT
run "variant 2 (pass 2: cursors / part_start / part_end addresses from the private segment)" $V/scratch_exp2.so --
run "variant 2, exchange_in_place=0" $V/scratch_exp2.so -- --option exchange_in_place=0
run "variant 3 (pass 1 on the low-priority partitioning stream: range-base address from the private segment)" $V/scratch_exp3.so --
run "variant 4 (K4p on the join stream: tuple / counter addresses from the private segment)" $V/scratch_exp4.so --
tail -5 $OUT
