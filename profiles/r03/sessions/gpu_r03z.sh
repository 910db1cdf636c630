#!/bin/bash
# round-3 GPU session Z: BatchNorm-backward column sums from the product's epilogue (bio stack) -- tests, A/B of the unprofiled bio step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "bio" > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
for k in 1 0 1 0; do
  echo "PGNN_BN_BWD_IN_GEMM=$k" >> $O/ab.txt
  PGNN_BN_BWD_IN_GEMM=$k timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
