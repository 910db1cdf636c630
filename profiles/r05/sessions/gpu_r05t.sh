#!/bin/bash
# round-5 GPU session T: the paired weight gradients with one workgroup per CU / fewer splits (less L2 pressure on the caller's stream)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05t
mkdir -p $O
cd $R
run() {
  echo "$1" >> $O/step_ab.txt
  env $1 timeout 300 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/step_ab.txt
}
for rep in 1 2; do
  run "PGNN_X=0"
  run "PGNN_DW_ONE_PER_CU=1"
  run "PGNN_DW_ONE_PER_CU=1 PGNN_DW_PAIR_WGS=256"
  run "PGNN_DW_PAIR_WGS=256"
  run "PGNN_DW_PAIR_WGS=384"
  run "PGNN_DW_PAIR_WGS=768"
  run "PGNN_FORK_VIA_LAUNCH=0"
done
cat $O/step_ab.txt
