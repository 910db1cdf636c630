#!/bin/bash
# round-4 GPU session I: A/B of the early per-layer fork event; schedule-equivalence test
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "side_stream_schedules or transposed_aggregation" > $O/tests_new.txt 2>&1
tail -n 5 $O/tests_new.txt
for k in "PGNN_X=0" "PGNN_FORK_EARLY=1" "PGNN_X=0" "PGNN_FORK_EARLY=1" "PGNN_X=0" "PGNN_FORK_EARLY=1"; do
  echo "$k" >> $O/ab.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
