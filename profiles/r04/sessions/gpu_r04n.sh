#!/bin/bash
# round-4 GPU session N: the head's own gradients deferred to an auxiliary stream: tests, A/B, suite
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04n
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference.py -m gpu -q -x -k "deferred_head or hip_graph or masking or epoch_accuracy" > $O/tests_new.txt 2>&1
tail -n 5 $O/tests_new.txt
for k in "PGNN_X=0" "PGNN_DEFER_HEAD_GRADS=0" "PGNN_X=0" "PGNN_DEFER_HEAD_GRADS=0" "PGNN_X=0" "PGNN_DEFER_HEAD_GRADS=0"; do
  echo "$k" >> $O/ab.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
timeout 300 python -m pytest tests/test_gpu_parallel.py -m gpu -q > $O/tests_parallel.txt 2>&1
tail -n 3 $O/tests_parallel.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_ops.py::test_products_on_two_fp16_planes_against_float64 --deselect tests/test_gpu_parallel.py > $O/tests_all.txt 2>&1
tail -n 5 $O/tests_all.txt
