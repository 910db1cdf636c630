#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
timeout 600 python tools/host_profile.py 200 > $O/host_profile.txt 2>&1
tail -5 $O/pytest.log; head -60 $O/host_profile.txt
