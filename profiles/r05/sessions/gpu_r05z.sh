#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05z
mkdir -p $O
cd $R
run() {
  echo "$1" >> $O/ctx_ab.txt
  env $1 timeout 300 python tools/ctx_host_profile.py 200 2>/dev/null | grep "^step" >> $O/ctx_ab.txt
}
for rep in 1 2; do
  run "PGNN_CTX_PIPELINE=1"
  run "PGNN_CTX_PIPELINE=0"
  run "PGNN_CTX_PIPELINE=1 PGNN_SIDE_STREAM=0"
  run "PGNN_CTX_PIPELINE=0 PGNN_SIDE_STREAM=0"
done
cat $O/ctx_ab.txt
