#!/bin/bash
# round-5 GPU session L: the launch layer (every launch through csrc/launch.h) and PGNN_CALL_GRAPHS=1: the new test, the whole
# suite as shipped (graphs off), the chem suites with graphs ON, the unchanged script's phases with and without
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "call_graphs" > $O/tests_graphs.txt 2>&1
tail -n 12 $O/tests_graphs.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/tests_all.txt 2>&1
tail -n 3 $O/tests_all.txt
PGNN_CALL_GRAPHS=1 timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference.py -m gpu -q -k "chem or masking or fused or gin" > $O/tests_graphs_on.txt 2>&1
tail -n 6 $O/tests_graphs_on.txt
for g in 0 1; do
  echo "PGNN_CALL_GRAPHS=$g" | tee -a $O/script_phases.txt
  PGNN_CALL_GRAPHS=$g timeout 300 python tools/script_phases.py 256 300 2>&1 | tail -n 2 | tee -a $O/script_phases.txt
done
for g in 0 1; do
  echo "PGNN_CALL_GRAPHS=$g step" | tee -a $O/step.txt
  PGNN_CALL_GRAPHS=$g timeout 300 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 | tee -a $O/step.txt
done
