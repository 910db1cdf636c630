#!/bin/bash
# round-5 GPU session E: whole GPU suite with the fused mlp wired into the chem stack, then the 16 384-graph step fused / unfused
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $O/tests_all.txt 2>&1
tail -n 6 $O/tests_all.txt
for k in "PGNN_MLP_FUSED=1" "PGNN_MLP_FUSED=0" "PGNN_MLP_FUSED=1" "PGNN_MLP_FUSED=0"; do
  echo "graphs 16384 $k" | tee -a $O/ab.txt
  env $k timeout 300 python tools/step_profile.py 16384 12 3 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
done
for k in "PGNN_MLP_FUSED=1" "PGNN_MLP_FUSED=0"; do
  echo "graphs 2048 $k" | tee -a $O/ab.txt
  env $k timeout 300 python tools/step_profile.py 2048 30 5 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
done
