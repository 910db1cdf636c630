#!/bin/bash
# round-5 GPU session I: multi-GPU readiness on one GPU -- the torchrun bench smoke as a test, the milestone test, a 200-step two-rank soak
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parallel.py -m gpu -q -x > $O/tests_parallel.txt 2>&1
tail -n 5 $O/tests_parallel.txt
timeout 600 python tools/soak_two_ranks.py 200 > $O/soak_two_ranks.txt 2>&1
echo "soak rc=$?" | tee -a $O/soak_two_ranks.txt
tail -n 6 $O/soak_two_ranks.txt
