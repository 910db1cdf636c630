#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ag
mkdir -p $O
cd $R
PGNN_DW_WAVES=4 timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "bond_table or side_stream or batchnorm_backward_sums" > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
run() {
  echo "$1" >> $O/step_ab.txt
  env $1 timeout 300 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/step_ab.txt
}
for rep in 1 2 3; do
  run "PGNN_DW_WAVES=8"
  run "PGNN_DW_WAVES=4"
done
cat $O/step_ab.txt
cd /tmp && export TMPDIR=/tmp
name=step_b256_w4
PGNN_DW_WAVES=4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
rm -rf $O/prof_$name
python $R/tools/kstats.py $O/${name}_kernel_stats.csv 10
