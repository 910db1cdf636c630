#!/bin/bash
# round-4 GPU session R: k_gemm2pr (weight planes resident in LDS): bit-identity against the tiled kernel, large-M timings
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04r
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "resident_plane" > $O/tests_new.txt 2>&1
tail -n 12 $O/tests_new.txt
timeout 300 python tools/gemm2p_large.py > $O/large.txt 2>&1
cat $O/large.txt
