#!/bin/bash
# step times on a fresh box (box-to-box check)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04z
mkdir -p $O
cd $R
(cat /proc/cpuinfo | grep "model name" | head -1; nproc) | tee $O/host.txt
for i in 1 2 3; do timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt; done
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 | tee -a $O/ab.txt
timeout 100 python tools/ctx_step_profile.py 256 100 2>&1 | tail -n 1 | tee -a $O/ab.txt
