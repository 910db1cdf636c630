#!/bin/bash
# round-4 GPU session K: A/B of 128x160 tiles for the paired weight gradients
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
PGNN_DW_TILE_M=128 timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "weight_gradient_pair" > $O/tests_new.txt 2>&1
tail -n 5 $O/tests_new.txt
for k in "PGNN_X=0" "PGNN_DW_TILE_M=128" "PGNN_X=0" "PGNN_DW_TILE_M=128"; do
  echo "$k" >> $O/ab.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
for k in "PGNN_X=0" "PGNN_DW_TILE_M=128"; do
  echo "bio $k" >> $O/ab.txt
  env $k timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
