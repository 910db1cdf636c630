#!/bin/bash
# round-5 GPU session H: fused mlp with H stored as whole 64-byte runs (two permlane swaps per register): tests, timing
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused_mlp" > $O/tests_fused.txt 2>&1
tail -n 4 $O/tests_fused.txt
timeout 300 python tools/mlp_fused_bench.py 262144 65536 32768 --iters 20 --out $O/mlp_fused_ab.jsonl > $O/bench.txt 2>&1
tail -n 4 $O/bench.txt
