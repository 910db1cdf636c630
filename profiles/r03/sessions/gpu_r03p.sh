#!/bin/bash
# round-3 GPU session P: weight-plane products at every size: large-batch tests, large-batch step with / without
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -k "large or planes or one_call or chem" > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
timeout 300 python tools/step_profile.py 4096 20 3 epoch 2>&1 | tail -n 1
PGNN_GEMM_WP=0 timeout 300 python tools/step_profile.py 4096 20 3 epoch 2>&1 | tail -n 1
timeout 300 python tools/step_profile.py 16384 10 2 epoch 2>&1 | tail -n 1
PGNN_GEMM_WP=0 timeout 300 python tools/step_profile.py 16384 10 2 epoch 2>&1 | tail -n 1
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1
