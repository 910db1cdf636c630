#!/bin/bash
# round-4 GPU session AA: gradient milestone + overlapped all-reduce: the parallel tests, the C-ABI export tests, step sanity
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04aa
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parallel.py -m gpu -q -x > $O/tests_parallel.txt 2>&1
tail -n 12 $O/tests_parallel.txt
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
