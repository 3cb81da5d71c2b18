cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_prepartitioned.py -q -k another_stream > gpurun_out/r04_two_stream_test_product.txt 2>&1; echo "product rc=$?"
HJGPU_LIBRARY=$PWD/hash_join_codes_knl_amd/lib/variants/scratch_exp9.so python -m pytest tests/test_gpu_prepartitioned.py -q -k another_stream > gpurun_out/r04_two_stream_test_variant9.txt 2>&1; echo "variant9 rc=$?"
for s in "256000000 1000000000" "512000000 1000000000" "1000000000 1000000000" "1000000000 4000000000"; do
  echo "== $s" >> gpurun_out/r04_big_build.txt
  timeout -k 10 300 python tools/big_case.py $s >> gpurun_out/r04_big_build.txt 2>&1 || echo "big_case $s rc=$?"
done
bash tools/r04_placement_dist.sh; echo "dist rc=$?"
tail -6 gpurun_out/r04_placement_dist.txt
