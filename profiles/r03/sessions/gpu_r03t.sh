#!/bin/bash
# round-3 GPU session T: widened single-block scan (graph / grouping tests), host profile of the bio step, unprofiled steps
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03t
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "graph or group_by or edge_head or embed or pool" > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 > $O/steps.txt
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/steps.txt
cat $O/steps.txt
timeout 200 python tools/bio_host_profile.py 256 60 > $O/bio_host_profile.txt 2>&1
head -n 70 $O/bio_host_profile.txt
