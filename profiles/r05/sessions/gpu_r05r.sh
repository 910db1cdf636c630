#!/bin/bash
# round-5 GPU session R: bond-table gradients out of the dW1 product, second version (the G^T W1 launch with eight rows' loads in
# flight): test, alternating A/B of the 256-graph step, kernel statistics of the step with and without
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05r
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "bond_table or side_stream" > $O/tests_bond.txt 2>&1
tail -n 3 $O/tests_bond.txt
for rep in 1 2 3; do
  for flag in 1 0; do
    echo "PGNN_BOND_IN_DW=$flag" >> $O/step_ab.txt
    PGNN_BOND_IN_DW=$flag timeout 300 python tools/step_profile.py 256 300 20 epoch 2>/dev/null | tail -n 1 >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
cd /tmp && export TMPDIR=/tmp
for flag in 1 0; do
  name=step_b256_bond$flag
  PGNN_BOND_IN_DW=$flag timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
  gzip -f $O/${name}_trace.csv
  rm -rf $O/prof_$name
  python $R/tools/kstats.py $O/${name}_kernel_stats.csv 16
  tail -n 1 $O/$name.log | cut -c1-300
done
