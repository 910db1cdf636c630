#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
timeout 600 python bench.py --no-extra-configs --no-cpu-baseline --no-roofline > $O/bench.json 2>$O/bench.err
timeout 600 python tools/host_profile.py 200 > $O/host_profile.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail; cat $O/bench.json; head -8 $O/host_profile.txt | tail -3
