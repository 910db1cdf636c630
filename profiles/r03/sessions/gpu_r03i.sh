#!/bin/bash
# round-3 GPU session I: persistent k_gemm3w (cross-tile prefetch): bit equality + times per shape, op tests
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
timeout 400 tools/bin/gemm3w_bench 6747 10249 41269 > $O/gemm3w_bench.txt 2>&1
grep -E "rows|k_gemm3 |cfg|EQUAL|DIFF" $O/gemm3w_bench.txt | grep -v phases | head -40
PGNN_GEMM3W_PERSIST=0 timeout 400 tools/bin/gemm3w_bench 6747 10249 41269 > $O/gemm3w_bench_nopersist.txt 2>&1
grep -E "rows|cfg -1" $O/gemm3w_bench_nopersist.txt | head -10
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "linear or planes or gemm or split" > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
