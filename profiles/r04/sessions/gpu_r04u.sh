#!/bin/bash
# round-4 GPU session U: k_gemm2pr after a change: bit-identity, the products' tests, large-M and 10 k-row timings, bio / chem steps
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04u
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -k "resident or two_fp16 or two_plane or planes or one_call or bit" > $O/tests_new.txt 2>&1
tail -n 3 $O/tests_new.txt
timeout 300 python tools/gemm2p_large.py 262144 65536 10249 > $O/large.txt 2>&1
cat $O/large.txt
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 | tee -a $O/ab.txt
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
timeout 200 python tools/step_profile.py 16384 12 3 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
