cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  HJGPU_PLACEMENT_LOG=1 timeout -k 5 120 python tools/alloc_luck.py serial 1 phj 2>&1 | grep 'context\|placement' | sed "s/^/process $i: /"
done
