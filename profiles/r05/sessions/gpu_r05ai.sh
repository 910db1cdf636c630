#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ai
mkdir -p $O
cd $R
timeout 900 python -X faulthandler -m pytest tests/test_gpu_models.py tests/test_gpu_parallel.py -m gpu -q -x > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
run() {
  echo "$1" >> $O/ab.txt
  env $1 timeout 300 python tools/ctx_host_profile.py 200 2>/dev/null | grep "^step" >> $O/ab.txt
  env $1 timeout 300 python tools/chem_host_profile.py 300 2>/dev/null | grep "^step" >> $O/ab.txt
}
for rep in 1 2; do
  run "PGNN_AUX_WORKER=1"
  run "PGNN_AUX_WORKER=0"
done
cat $O/ab.txt
