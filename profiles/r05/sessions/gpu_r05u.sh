#!/bin/bash
# round-5 GPU session U: PGNN_FORK_LATE=1 -- a layer's parameter gradients forked behind the NEXT layer's BatchNorm elementwise pass
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05u
mkdir -p $O
cd $R
PGNN_FORK_LATE=1 timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "bond_table or side_stream or batchnorm_backward_sums or one_call" > $O/tests_late.txt 2>&1
tail -n 3 $O/tests_late.txt
run() {
  echo "$1" >> $O/step_ab.txt
  env $1 timeout 300 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/step_ab.txt
}
for rep in 1 2 3; do
  run "PGNN_X=0"
  run "PGNN_FORK_LATE=1"
done
cat $O/step_ab.txt
cd /tmp && export TMPDIR=/tmp
name=step_b256_forklate
PGNN_FORK_LATE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
gzip -f $O/${name}_trace.csv
rm -rf $O/prof_$name
tail -n 1 $O/$name.log | cut -c1-300
