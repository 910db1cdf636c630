#!/bin/bash
# round-4 GPU session Y: full GPU suite with the resident-plane thresholds from the sweep; bio / chem / ctx step times
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04y
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests_all.txt 2>&1
tail -n 5 $O/tests_all.txt
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 | tee -a $O/ab.txt
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 | tee -a $O/ab.txt
timeout 100 python tools/ctx_step_profile.py 256 100 2>&1 | tail -n 1 | tee -a $O/ab.txt
