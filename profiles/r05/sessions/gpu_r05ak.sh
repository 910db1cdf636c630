#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ak
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_loader.py tests/test_gpu_reference.py -m gpu -q -x -k "substruct or context" > $O/tests.txt 2>&1
tail -n 2 $O/tests.txt
for rep in 1 2 3; do
for v in 1 0; do
  echo "PGNN_CTX_PLAN_STREAM=$v" >> $O/ab.txt
  PGNN_CTX_PLAN_STREAM=$v timeout 300 python tools/ctx_host_profile.py 200 2>/dev/null | grep "^step" >> $O/ab.txt
done
done
cat $O/ab.txt
