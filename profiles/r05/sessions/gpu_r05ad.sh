#!/bin/bash
# round-5 GPU session AD: the paired weight gradients with two LDS stages (PGNN_DW_DB=1): tests, step A/B, kernel statistics
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ad
mkdir -p $O
cd $R
PGNN_DW_DB=1 timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "bond_table or side_stream or batchnorm_backward_sums or one_call_network_on_weight" > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
run() {
  echo "$1" >> $O/step_ab.txt
  env $1 timeout 300 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/step_ab.txt
}
for rep in 1 2 3; do
  run "PGNN_DW_DB=0"
  run "PGNN_DW_DB=1"
done
cat $O/step_ab.txt
cd /tmp && export TMPDIR=/tmp
for flag in 1 0; do
name=step_b256_db$flag
PGNN_DW_DB=$flag timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
rm -rf $O/prof_$name
python $R/tools/kstats.py $O/${name}_kernel_stats.csv 10
done
