#!/bin/bash
# round-4 GPU session Q: ticket-fold stress test under memory pressure; bio edge prediction / Deep Graph Infomax against the
# reference's trajectories; step-time sanity
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04q
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_reference.py -m gpu -q -k "memory_pressure or edgepred or infomax" > $O/tests_new.txt 2>&1
tail -n 15 $O/tests_new.txt
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
cat $O/ab.txt
