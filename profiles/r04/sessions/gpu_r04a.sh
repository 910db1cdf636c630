#!/bin/bash
# round-4 GPU session A: ADVICE r03 fixes (publish_commit) -- full suite; the same suite with PGNN_GEMM_2P=1; step profile with the knob on
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q > $O/tests_default.txt 2>&1
tail -n 3 $O/tests_default.txt
PGNN_GEMM_2P=1 timeout 600 python -m pytest tests -m gpu -q > $O/tests_2p.txt 2>&1
tail -n 15 $O/tests_2p.txt
cd /tmp && export TMPDIR=/tmp
name=step_b256_2p
PGNN_GEMM_2P=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
rm -rf $O/prof_$name
cd $R
python tools/kstats.py $O/${name}_kernel_stats.csv 45 > $O/${name}_kstats.txt
python tools/step_timeline.py $O/${name}_trace.csv > $O/${name}_timeline.txt 2>&1
gzip -f $O/*_trace.csv
cat $O/${name}_kstats.txt
for k in 1 0; do
  echo "PGNN_GEMM_2P=$k" >> $O/ab.txt
  PGNN_GEMM_2P=$k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
