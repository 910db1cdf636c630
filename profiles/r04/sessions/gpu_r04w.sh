#!/bin/bash
# round-4 GPU session W: the resident-plane kernel forced onto the 256-graph batches (bio 10 249 rows, chem 6 740, ctx)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04w
mkdir -p $O
cd $R
for k in 1 2 1 2; do
  echo "bio PGNN_GEMM2P_RES=$k" >> $O/ab.txt
  PGNN_GEMM2P_RES=$k timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
done
for k in 1 2 1 2; do
  echo "chem PGNN_GEMM2P_RES=$k" >> $O/ab.txt
  PGNN_GEMM2P_RES=$k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
