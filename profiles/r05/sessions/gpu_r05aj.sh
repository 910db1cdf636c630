#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05aj
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_loader.py -m gpu -q -x -k "substruct" > $O/tests.txt 2>&1
tail -n 2 $O/tests.txt
PGNN_DW_PAIR_MIN_ROWS=512 timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference.py -m gpu -q -x -k "bond_table or contextpred or one_call_network_on" > $O/tests2.txt 2>&1
tail -n 2 $O/tests2.txt
for rep in 1 2 3; do
for v in 2048 1024 512; do
  echo "PGNN_DW_PAIR_MIN_ROWS=$v" >> $O/ab.txt
  PGNN_DW_PAIR_MIN_ROWS=$v timeout 300 python tools/ctx_step_profile.py 256 100 2>/dev/null | tail -n 1 >> $O/ab.txt
done
done
cat $O/ab.txt
