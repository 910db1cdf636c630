#!/bin/bash
# round-3 GPU session M: far-row prefetch in the chem aggregation kernel: bit-exactness with the knob on / off, robustness leg A/B, step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "aggregate" > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
PGNN_DMA_PF=1 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -k "aggregate or chem or one_call" > $O/tests_pf.txt 2>&1
tail -n 3 $O/tests_pf.txt
timeout 600 python tools/agg_robust_ab.py > $O/agg_robust_ab.txt 2>&1
cat $O/agg_robust_ab.txt | grep PGNN
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1
PGNN_DMA_PF=1 timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1
