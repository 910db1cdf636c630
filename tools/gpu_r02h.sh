#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python $R/tools/step_profile.py 256 30 5 > $O/step.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/step_b256_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/step_b256_kernel_trace.csv
rm -rf $O/prof
tail -2 $O/step.log; wc -l $O/step_b256_kernel_trace.csv
