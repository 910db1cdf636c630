#!/bin/bash
# round-5 GPU session M: PGNN_CALL_GRAPHS=1 with the capture on a private stream: the new test, chem suites with graphs ON, script phases, step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05m
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "call_graphs" > $O/tests_graphs.txt 2>&1
tail -n 12 $O/tests_graphs.txt
PGNN_CALL_GRAPHS=1 timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference.py tests/test_gpu_parallel.py -m gpu -q > $O/tests_graphs_on.txt 2>&1
tail -n 8 $O/tests_graphs_on.txt
for g in 0 1; do
  echo "PGNN_CALL_GRAPHS=$g" | tee -a $O/script_phases.txt
  PGNN_CALL_GRAPHS=$g timeout 300 python tools/script_phases.py 256 300 2>&1 | tail -n 2 | tee -a $O/script_phases.txt
done
for g in 0 1; do
  echo "PGNN_CALL_GRAPHS=$g step" | tee -a $O/step.txt
  PGNN_CALL_GRAPHS=$g timeout 300 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 | tee -a $O/step.txt
  PGNN_CALL_GRAPHS=$g timeout 300 python tools/step_profile.py 256 300 20 end 2>&1 | tail -n 1 | tee -a $O/step.txt
done
