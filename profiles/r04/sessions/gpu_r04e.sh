#!/bin/bash
# round-4 GPU session E: both weight gradients of a layer in one launch (k_gemm3_pair) -- op test, stack tests, A/B of the step; reciprocal in the folds
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -k "weight_gradient_pair or folded_inside or one_call or transposed_aggregation or linear" > $O/tests_new.txt 2>&1
tail -n 8 $O/tests_new.txt
for k in "PGNN_DW_PAIR=1" "PGNN_DW_PAIR=0" "PGNN_DW_PAIR=1" "PGNN_DW_PAIR=0" "PGNN_BN_STATS_FOLD=0"; do
  echo "$k" >> $O/ab.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
for k in "PGNN_DW_PAIR=1" "PGNN_DW_PAIR=0"; do
  echo "bio $k" >> $O/ab.txt
  env $k timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
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
