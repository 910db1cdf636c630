#!/bin/bash
# round-4 GPU session V: kernel statistics of the 16 384-graph step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04v
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
name=step_b16384
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 16384 8 2 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
rm -rf $O/prof_$name
cd $R
python tools/kstats.py $O/${name}_kernel_stats.csv 40 > $O/${name}_kstats.txt
cat $O/${name}_kstats.txt
tail -n 2 $O/$name.log
