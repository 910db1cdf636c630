#!/bin/bash
# round-4 GPU session T: k_gemm2pr as shipped (thresholds 16 384 / 32 768 rows): product tests, large-M timings, step sanity, large batches
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04t
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -k "resident or two_fp16 or two_plane or planes or one_call or bit" > $O/tests_new.txt 2>&1
tail -n 6 $O/tests_new.txt
timeout 300 python tools/gemm2p_large.py > $O/large.txt 2>&1
cat $O/large.txt
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
for g in 2048 16384; do
  for k in 0 1; do
    echo "graphs $g PGNN_GEMM2P_RES=$k" | tee -a $O/ab.txt
    PGNN_GEMM2P_RES=$k timeout 200 python tools/step_profile.py $g 12 3 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
  done
done
