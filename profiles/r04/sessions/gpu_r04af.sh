#!/bin/bash
# round-4 GPU session AF: BatchNorm-backward sums from the transposed aggregation on the large batches' (non-temporal) instance
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04af
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "transposed_aggregation or side_stream or bit or one_call" > $O/tests_new.txt 2>&1
tail -n 3 $O/tests_new.txt
for g in 16384 16384 4300; do
  for k in "PGNN_BN_BWD_IN_AGG=1" "PGNN_BN_BWD_IN_AGG=0"; do
    echo "graphs $g $k" | tee -a $O/ab.txt
    env $k timeout 200 python tools/step_profile.py $g 12 3 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
  done
done
