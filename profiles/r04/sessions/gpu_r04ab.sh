#!/bin/bash
# round-4 GPU session AB: the GPU suite with the resident-plane kernel forced onto every eligible product (PGNN_GEMM2P_RES=2), then as shipped
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04ab
mkdir -p $O
cd $R
PGNN_GEMM2P_RES=2 timeout 1500 python -m pytest tests -m gpu -q > $O/tests_res2.txt 2>&1
tail -n 8 $O/tests_res2.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_default.txt 2>&1
tail -n 3 $O/tests_default.txt
