#!/bin/bash
# round-4 GPU session U: k_gemm2pr variants: bit-identity + large-M timings
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04u
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "resident" > $O/tests_new.txt 2>&1
tail -n 3 $O/tests_new.txt
timeout 300 python tools/gemm2p_large.py 262144 65536 > $O/large.txt 2>&1
cat $O/large.txt
