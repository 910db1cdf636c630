#!/bin/bash
# round-2 GPU session Q: kernel mix and timeline of the bio masking step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bio -- python $R/tools/bio_step_profile.py 256 30 > $O/bio.log 2>&1
python $R/tools/kstats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) 45
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bio_step_kernel_stats.csv
cp $(find $O/prof -name "*kernel_trace.csv" | head -1) $O/bio_trace.csv
rm -rf $O/prof
cd $R
timeout 100 python tools/bio_step_profile.py 256 100 | tail -1
