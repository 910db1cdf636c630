#!/bin/bash
# round-5 GPU session S: the caller's stream of the chem backward without marker packets: fork[1] as the completion of the product's
# own dispatch (PGNN_FORK_VIA_LAUNCH=1), per-layer buffer sets (no lag waits), device-scope release on the fork / lag events
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s
mkdir -p $O
cd $R
PGNN_FORK_VIA_LAUNCH=1 timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "bond_table or side_stream or batchnorm_backward_sums" > $O/tests_fork.txt 2>&1
tail -n 3 $O/tests_fork.txt
run() {
  echo "$1" >> $O/step_ab.txt
  env $1 timeout 300 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/step_ab.txt
}
for rep in 1 2; do
  run "PGNN_X=0"
  run "PGNN_FORK_VIA_LAUNCH=1"
  run "PGNN_STACK_PER_LAYER_BUFFERS=1"
  run "PGNN_EVENT_DEVICE_RELEASE=1"
  run "PGNN_FORK_VIA_LAUNCH=1 PGNN_STACK_PER_LAYER_BUFFERS=1"
  run "PGNN_FORK_VIA_LAUNCH=1 PGNN_STACK_PER_LAYER_BUFFERS=1 PGNN_EVENT_DEVICE_RELEASE=1"
done
cat $O/step_ab.txt
cd /tmp && export TMPDIR=/tmp
name=step_b256_forklaunch
PGNN_FORK_VIA_LAUNCH=1 PGNN_STACK_PER_LAYER_BUFFERS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
gzip -f $O/${name}_trace.csv
rm -rf $O/prof_$name
tail -n 1 $O/$name.log | cut -c1-300
