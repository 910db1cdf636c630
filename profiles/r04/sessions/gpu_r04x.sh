#!/bin/bash
# round-4 GPU session X: where the resident-plane kernel overtakes the tiled one, per shape
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
timeout 600 python tools/gemm2p_sweep.py > $O/sweep.txt 2>&1
cat $O/sweep.txt
