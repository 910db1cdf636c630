#!/bin/bash
# round-4 GPU session P: transposed two-plane weight split through LDS tiles; sparse top-layer sums: tests, step times, profile
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -k "split_weights_2p or two_fp16 or two_plane or one_call or masked_rows_only" > $O/tests_new.txt 2>&1
tail -n 4 $O/tests_new.txt
for k in "PGNN_X=0" "PGNN_SPARSE_TOP_GRAD=0" "PGNN_X=0" "PGNN_SPARSE_TOP_GRAD=0" "PGNN_X=0"; do
  echo "$k" >> $O/ab.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
cat $O/ab.txt
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
grep -n "split2p\|bn_bwd_partial" $O/${name}_kstats.txt
sed -n 28,46p $O/${name}_timeline.txt
