#!/bin/bash
# round-3 GPU session B: the profiles behind the bench line -- roofline-only launches, the three step kernel mixes, full bench
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- "$@" > $O/$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
  rm -rf $O/prof_$name
  python $R/tools/kstats.py $O/${name}_kernel_stats.csv 45 > $O/${name}_kstats.txt
}
prof roofline_only python $R/bench.py --roofline-only
prof step_b256 python $R/tools/step_profile.py 256 30 5 epoch
prof bio_step python $R/tools/bio_step_profile.py 256 30
prof ctx_step python $R/tools/ctx_step_profile.py 256 30
cd $R
for n in step_b256 ctx_step bio_step; do python tools/step_timeline.py $O/${n}_trace.csv > $O/${n}_timeline.txt 2>&1; done
python tools/trace_gaps.py $O/step_b256_trace.csv > $O/step_b256_gaps.txt 2>&1
gzip -f $O/*_trace.csv
tail -n 1 $O/roofline_only.log > $O/roofline_only.json
python bench.py > $O/bench.json 2> $O/bench.err
cat $O/roofline_only_kstats.txt | head -12
tail -n 3 $O/step_b256.log | tail -n 1
