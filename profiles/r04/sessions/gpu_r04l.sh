#!/bin/bash
# round-4 GPU session L: 112x160 tiles (seven waves) for the two-plane products where 128-row tiles leave CUs empty: tests, A/B
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04l
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -k "two_fp16 or two_plane or one_call or side_stream_schedules or folded_inside" > $O/tests_new.txt 2>&1
tail -n 5 $O/tests_new.txt
for k in "PGNN_X=0" "PGNN_GEMM2P_T112=0" "PGNN_X=0" "PGNN_GEMM2P_T112=0" "PGNN_X=0" "PGNN_GEMM2P_T112=0"; do
  echo "$k" >> $O/ab.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
for k in "PGNN_X=0" "PGNN_GEMM2P_T112=0"; do
  echo "bio $k" >> $O/ab.txt
  env $k timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
