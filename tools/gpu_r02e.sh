#!/bin/bash
# round-2 GPU session E: native bio GAT tests, then the full GPU suite and the default bench
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_reference.py -q -p no:cacheprovider -k "gat or GAT" > $O/pytest_gat.log 2>&1
echo "pytest rc $?" >> $O/pytest_gat.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "pytest rc $?" >> $O/pytest_all.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
grep -E "passed|failed|^FAILED|^ERROR|rc " $O/pytest_gat.log | tail -15; grep -E "passed|failed|^FAILED|^ERROR|rc " $O/pytest_all.log | tail -15; cat $O/bench.json; tail -3 $O/bench.err
