#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail
