#!/bin/bash
# round-3 GPU session AB: two-plane products -- the parity test's measured deviations, and the reference-fixture deviations with the knob on
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03ab
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_gpu_models.py -m gpu -q -s -k "two_fp16" > $O/tests.txt 2>&1
grep -E "two planes vs|passed|failed" $O/tests.txt
rm -f gpurun_out/parity_metrics.jsonl
PGNN_GEMM_2P=1 timeout 300 python -m pytest tests/test_gpu_reference.py -m gpu -q > $O/ref_2p.txt 2>&1
tail -n 2 $O/ref_2p.txt
cp gpurun_out/parity_metrics.jsonl $O/parity_metrics_2p.jsonl
