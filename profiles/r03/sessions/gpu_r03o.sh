#!/bin/bash
# round-3 GPU session O: BatchNorm backward with its fold inside the partial launch: tests, step times with / without
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03o
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_reference.py -m gpu -x -q > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
for k in 1 0 1 0; do
  PGNN_BN_BWD_FOLD=$k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1
done
PGNN_BN_BWD_FOLD=1 timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1
PGNN_BN_BWD_FOLD=0 timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1
cd /tmp && export TMPDIR=/tmp
name=step_b256
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
rm -rf $O/prof_$name
cd $R
python tools/kstats.py $O/${name}_kernel_stats.csv 45 > $O/${name}_kstats.txt
python tools/step_timeline.py $O/${name}_trace.csv > $O/${name}_timeline.txt 2>&1
gzip -f $O/*_trace.csv
grep "bn_bwd" $O/${name}_kstats.txt
