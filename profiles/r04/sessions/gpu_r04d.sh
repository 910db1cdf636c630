#!/bin/bash
# round-4 GPU session D: BatchNorm statistics folded inside the second product's launch (forward) + the backward sums from the
# transposed aggregation with the loader wave in the tail's barriers; two-rank flow (faulthandler armed); A/B of the step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "transposed_aggregation or folded_inside or one_call or statistics_from" > $O/tests_new.txt 2>&1
tail -n 8 $O/tests_new.txt
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_parallel.py -m gpu -q -x > $O/tests_parallel_$i.txt 2>&1
  tail -n 3 $O/tests_parallel_$i.txt
done
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_ops.py::test_products_on_two_fp16_planes_against_float64 --deselect tests/test_gpu_parallel.py > $O/tests_all.txt 2>&1
tail -n 8 $O/tests_all.txt
for k in "PGNN_BN_STATS_FOLD=1" "PGNN_BN_STATS_FOLD=0" "PGNN_BN_STATS_FOLD=1" "PGNN_BN_STATS_FOLD=0 PGNN_BN_BWD_IN_AGG=0"; do
  echo "$k" >> $O/ab.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
name=step_b256
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
rm -rf $O/prof_$name
cd $R
python tools/kstats.py $O/${name}_kernel_stats.csv 14 > $O/${name}_kstats.txt
python tools/step_timeline.py $O/${name}_trace.csv > $O/${name}_timeline.txt 2>&1
gzip -f $O/*_trace.csv
cat $O/${name}_kstats.txt
