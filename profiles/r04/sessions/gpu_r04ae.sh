#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04ae
mkdir -p $O
cd $R
timeout 200 python tools/chem_host_profile.py 300 > $O/chem_host.txt 2>&1
cat $O/chem_host.txt
