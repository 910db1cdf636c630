#!/bin/bash
# round-3 GPU session R: single-workgroup graph build / grouping, bio payload by 16 lanes, encoder tables in the split launch,
# side-stream wait after the first BatchNorm backward -- full GPU suite, A/B of the unprofiled steps, timelines
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1
tail -n 5 $O/tests.txt
for k in 1 0; do
  echo "PGNN_GRAPH_SMALL=$k" >> $O/ab.txt
  PGNN_GRAPH_SMALL=$k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
for k in 1 0; do
  echo "PGNN_BIO_PAYLOAD16=$k PGNN_GROUP_SMALL=$k" >> $O/ab.txt
  PGNN_BIO_PAYLOAD16=$k PGNN_GROUP_SMALL=$k timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
done
timeout 100 python tools/ctx_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- "$@" > $O/$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
  rm -rf $O/prof_$name
  python $R/tools/kstats.py $O/${name}_kernel_stats.csv 45 > $O/${name}_kstats.txt
}
prof step_b256 python $R/tools/step_profile.py 256 30 5 epoch
prof bio_step python $R/tools/bio_step_profile.py 256 30
cd $R
for n in step_b256 bio_step; do python tools/step_timeline.py $O/${n}_trace.csv > $O/${n}_timeline.txt 2>&1; done
python tools/trace_gaps.py $O/step_b256_trace.csv > $O/step_b256_gaps.txt 2>&1
gzip -f $O/*_trace.csv
head -n 3 $O/step_b256_gaps.txt
grep -E "k_chem_graph_small|k_group_small|k_bio_payload|k_split_jobs" $O/step_b256_kstats.txt $O/bio_step_kstats.txt
