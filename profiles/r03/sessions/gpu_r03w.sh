#!/bin/bash
# round-3 GPU session W: where the caller's stream idles -- event placement / per-layer buffers / no side stream, unprofiled steps
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03w
mkdir -p $O
cd $R
run() { echo "$1" >> $O/ab.txt; env $1 timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt; }
run "PGNN_FORK_LATE=0"
run "PGNN_FORK_LATE=1"
run "PGNN_STACK_PER_LAYER_BUFFERS=1"
run "PGNN_STACK_PER_LAYER_BUFFERS=1 PGNN_FORK_LATE=1"
run "PGNN_FORK_LATE=0"
run "PGNN_FORK_LATE=1"
run "PGNN_SIDE_STREAM=0"
cat $O/ab.txt
