#!/bin/bash
# distribution of pass-1 time over FRESH PROCESSES for several `placement` settings
cd $GRAFT_REPO_ROOT
for p in 1 6 12 16; do
  for i in 1 2 3 4 5 6; do
    HJGPU_PLACEMENT=$p timeout -k 5 120 python tools/alloc_luck.py serial 1 phj | sed "s/^/placement=$p process $i: /"
  done
done
