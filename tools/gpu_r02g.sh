#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
timeout 600 python bench.py --no-extra-configs --no-cpu-baseline --no-roofline > $O/bench.json 2>$O/bench.err
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail; cat $O/bench.json
bash tools/gpu_r02h.sh > /dev/null 2>&1
