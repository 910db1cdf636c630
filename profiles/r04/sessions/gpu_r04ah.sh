#!/bin/bash
# round-4 GPU session AH: the masking head's forward at 16 rows per block for large batches: tests, 16 384-graph step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04ah
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "masked_head" > $O/tests_new.txt 2>&1
tail -n 3 $O/tests_new.txt
for k in "PGNN_HEAD_ROWS16_FROM=16384" "PGNN_HEAD_ROWS16_FROM=100000000" "PGNN_HEAD_ROWS16_FROM=16384" "PGNN_HEAD_ROWS16_FROM=100000000"; do
  echo "graphs 16384 $k" | tee -a $O/ab.txt
  env $k timeout 200 python tools/step_profile.py 16384 12 3 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
done
