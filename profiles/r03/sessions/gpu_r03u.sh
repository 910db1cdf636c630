#!/bin/bash
# round-3 GPU session U: head soft-max by wave shuffles, loader id staging -- their tests, the unprofiled steps, head kernel durations
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03u
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_loader.py -m gpu -x -q -k "head or loader or epoch" > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 > $O/steps.txt
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/steps.txt
timeout 200 python tools/bio_host_profile.py 256 60 2>&1 | grep "loader in the loop" >> $O/steps.txt
cat $O/steps.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python $R/tools/step_profile.py 256 30 5 epoch > $O/step.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/step_kernel_stats.csv; rm -rf $O/prof
python $R/tools/kstats.py $O/step_kernel_stats.csv 60 | grep -E "head|scan_single|fillBuffer"
