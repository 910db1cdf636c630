#!/bin/bash
# round-5 GPU session Q: bond-table gradients out of the dW1 product (PGNN_BOND_IN_DW): its test, the model / reference suites,
# then alternating A/B of the 256-graph chem step and the context-prediction step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05q
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "bond_table or side_stream or one_call_network_equals or batchnorm_backward_sums" > $O/tests_bond.txt 2>&1
tail -n 5 $O/tests_bond.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference.py tests/test_gpu_parallel.py -m gpu -q > $O/tests_models.txt 2>&1
tail -n 5 $O/tests_models.txt
for rep in 1 2 3; do
  for flag in 1 0; do
    echo "PGNN_BOND_IN_DW=$flag" >> $O/step_ab.txt
    PGNN_BOND_IN_DW=$flag timeout 300 python tools/step_profile.py 256 300 20 epoch 2>/dev/null | tail -n 1 >> $O/step_ab.txt
  done
done
for rep in 1 2; do
  for flag in 1 0; do
    echo "PGNN_BOND_IN_DW=$flag" >> $O/ctx_ab.txt
    PGNN_BOND_IN_DW=$flag timeout 300 python tools/ctx_step_profile.py 256 100 2>/dev/null | tail -n 1 >> $O/ctx_ab.txt
  done
done
cat $O/step_ab.txt $O/ctx_ab.txt
