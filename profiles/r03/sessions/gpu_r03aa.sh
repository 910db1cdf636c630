#!/bin/bash
# round-3 GPU session AA: the stacks' products on two fp16 planes (PGNN_GEMM_2P) -- parity test, the default-path stack tests, A/B of the steps
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03aa
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "two_fp16 or one_call" > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
for k in 1 0 1 0; do
  echo "PGNN_GEMM_2P=$k" >> $O/ab.txt
  PGNN_GEMM_2P=$k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
for k in 1 0; do
  echo "PGNN_GEMM_2P=$k" >> $O/ab.txt
  PGNN_GEMM_2P=$k timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
