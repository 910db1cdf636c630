#!/bin/bash
# round-2 GPU session M: epoch read-back + trimmed launches: targeted tests, then the 256-graph step in both read-back modes
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -p no:cacheprovider -k "head or adam or epoch or masking or graph or group or embed or loader or one_call" 2>&1 | tail -6
for rb in epoch end; do timeout 100 python tools/step_profile.py 256 300 20 $rb 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python $R/tools/step_profile.py 256 30 5 epoch > $O/step.log 2>&1
python $R/tools/kstats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) 70 | awk '{c+=$1} END {print "kernel launches in 35 steps:", c, "=", c/35, "per step"}'
cp $(find $O/prof -name "*kernel_trace.csv" | head -1) $O/step_trace.csv
rm -rf $O/prof
