#!/bin/bash
# round-4 GPU session G: A/B of the side stream's priority and of the BatchNorm-backward apply grid
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
for k in "PGNN_X=0" "PGNN_SIDE_PRIORITY=1" "PGNN_BN_APPLY_BPC=2" "PGNN_BN_APPLY_BPC=4" "PGNN_X=0" "PGNN_SIDE_PRIORITY=1" "PGNN_SIDE_PRIORITY=1 PGNN_BN_APPLY_BPC=4"; do
  echo "$k" >> $O/ab.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
for k in "PGNN_X=0" "PGNN_SIDE_PRIORITY=1"; do
  echo "bio $k" >> $O/ab.txt
  env $k timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
