#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05am
mkdir -p $O
cd $R
PGNN_LAG_WAIT_EARLY=1 timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "side_stream or bond_table or batchnorm_backward_sums" > $O/tests.txt 2>&1
tail -n 2 $O/tests.txt
for rep in 1 2 3; do
for v in 0 1; do
  echo "PGNN_LAG_WAIT_EARLY=$v" >> $O/ab.txt
  PGNN_LAG_WAIT_EARLY=$v timeout 300 python tools/step_profile.py 256 300 20 epoch 2>/dev/null | tail -n 1 >> $O/ab.txt
done
done
cat $O/ab.txt
