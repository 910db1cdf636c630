#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/ctx_step_profile.py 256 30 > $O/ctx_plain.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ctx -- python $R/tools/ctx_step_profile.py 256 30 > $O/ctx_step.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/ctx_step_kernel_stats.csv
rm -rf $O/prof
tail -1 $O/ctx_plain.log; tail -1 $O/ctx_step.log
