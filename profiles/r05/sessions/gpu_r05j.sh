#!/bin/bash
# round-5 GPU session J: bio stack with both weight gradients of a layer in one launch (k_gemm3_pair): tests, step A/B
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference.py -m gpu -q -x -k "bio" > $O/tests_bio.txt 2>&1
tail -n 4 $O/tests_bio.txt
for k in 1 0 1 0; do
  echo "PGNN_DW_PAIR=$k" | tee -a $O/ab.txt
  PGNN_DW_PAIR=$k timeout 200 python tools/bio_step_profile.py 256 60 2>&1 | tail -n 2 | tee -a $O/ab.txt
done
