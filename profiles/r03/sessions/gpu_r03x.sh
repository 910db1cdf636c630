#!/bin/bash
# round-3 GPU session X: structure build one batch ahead on the prefetch stream -- its test, A/B of the unprofiled steps, a timeline
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03x
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_loader.py -m gpu -x -q -k "epoch or ahead or loader" > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
for a in 1 0 1 0; do timeout 100 python tools/step_profile.py 256 300 20 epoch $a 2>&1 | tail -n 1 >> $O/ab.txt; done
for a in 1 0; do timeout 100 python tools/bio_step_profile.py 256 100 $a 2>&1 | tail -n 1 >> $O/ab.txt; echo "  (bio ahead=$a)" >> $O/ab.txt; done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python $R/tools/step_profile.py 256 30 5 epoch > $O/step.log 2>&1
cp $(find $O/prof -name "*kernel_trace.csv" | head -1) $O/step_b256_trace.csv; rm -rf $O/prof
cd $R
python tools/step_timeline.py $O/step_b256_trace.csv > $O/step_b256_timeline.txt 2>&1
python tools/trace_gaps.py $O/step_b256_trace.csv > $O/step_b256_gaps.txt 2>&1
gzip -f $O/step_b256_trace.csv
head -n 3 $O/step_b256_gaps.txt
