#!/bin/bash
# round-2 GPU session O: BatchNorm statistics from the GEMM epilogue: tests, then the 256-graph step with and without
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -p no:cacheprovider -k "colstats or batchnorm or linear or one_call or statistics or masking or epoch or determin" 2>&1 | tail -6
for f in 1 0; do PGNN_BN_STATS_IN_GEMM=$f timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -1; done
