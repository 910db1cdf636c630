#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05v
mkdir -p $O
cd $R
run() {
  echo "$1" >> $O/step_ab.txt
  env $1 timeout 300 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/step_ab.txt
}
for rep in 1 2 3; do
  run "PGNN_DW_SPLIT_MODE=0"
  run "PGNN_DW_SPLIT_MODE=1"
done
cat $O/step_ab.txt
for m in 0 1; do
echo "ctx PGNN_DW_SPLIT_MODE=$m" >> $O/ctx_ab.txt
PGNN_DW_SPLIT_MODE=$m timeout 300 python tools/ctx_step_profile.py 256 100 2>/dev/null | tail -n 1 >> $O/ctx_ab.txt
done
cat $O/ctx_ab.txt
