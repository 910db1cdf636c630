#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
name=ctx_step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/ctx_step_profile.py 256 33 > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
gzip -f $O/${name}_trace.csv
rm -rf $O/prof_$name
tail -n 1 $O/$name.log | cut -c1-300
cd $R
timeout 300 python tools/chem_host_profile.py > $O/chem_host.txt 2>&1
tail -n 30 $O/chem_host.txt
