#!/bin/bash
# round-4 GPU session J: row maxima of agg / dz from their producers (aggregation kernel, BatchNorm-backward apply): tests, A/B, suite
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "side_stream_schedules or transposed_aggregation or one_call or folded_inside" > $O/tests_new.txt 2>&1
tail -n 5 $O/tests_new.txt
for k in "PGNN_X=0" "PGNN_PRODUCER_AMAX=0" "PGNN_X=0" "PGNN_PRODUCER_AMAX=0" "PGNN_X=0" "PGNN_PRODUCER_AMAX=0"; do
  echo "$k" >> $O/ab.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_ops.py::test_products_on_two_fp16_planes_against_float64 > $O/tests_all.txt 2>&1
tail -n 5 $O/tests_all.txt
