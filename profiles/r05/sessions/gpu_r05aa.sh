#!/bin/bash
# round-5 GPU session AA: side stream only from 5 000 rows on, pipelined substructure/context loader, cached module references
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05aa
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
run() {
  echo "$1" >> $O/ctx_ab.txt
  env $1 timeout 300 python tools/ctx_host_profile.py 200 2>/dev/null | grep "^step" >> $O/ctx_ab.txt
}
for rep in 1 2; do
  run "PGNN_X=0"
  run "PGNN_SIDE_MIN_ROWS=0"
  run "PGNN_CTX_PIPELINE=0"
  run "PGNN_SIDE_MIN_ROWS=0 PGNN_CTX_PIPELINE=0"
done
cat $O/ctx_ab.txt
for i in 1 2; do timeout 300 python tools/step_profile.py 256 300 20 epoch 2>/dev/null | tail -n 1; done | tee $O/chem.txt
timeout 300 python tools/chem_host_profile.py 300 2>/dev/null | grep "^step" | tee -a $O/chem.txt
