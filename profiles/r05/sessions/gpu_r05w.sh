#!/bin/bash
# round-5 GPU session W: one-round splits of the 64x160 weight-gradient products as the default: op + model tests, chem / bio / ctx A/B
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05w
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
for rep in 1 2; do
for m in 1 0; do
echo "bio PGNN_DW_SPLIT_MODE=$m" >> $O/ab.txt
PGNN_DW_SPLIT_MODE=$m timeout 300 python tools/bio_step_profile.py 256 60 2>/dev/null | tail -n 1 >> $O/ab.txt
echo "ctx PGNN_DW_SPLIT_MODE=$m" >> $O/ab.txt
PGNN_DW_SPLIT_MODE=$m timeout 300 python tools/ctx_step_profile.py 256 100 2>/dev/null | tail -n 1 >> $O/ab.txt
echo "chem PGNN_DW_SPLIT_MODE=$m" >> $O/ab.txt
PGNN_DW_SPLIT_MODE=$m timeout 300 python tools/step_profile.py 256 300 20 epoch 2>/dev/null | tail -n 1 >> $O/ab.txt
done
done
cat $O/ab.txt
