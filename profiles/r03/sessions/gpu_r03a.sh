#!/bin/bash
# round-3 GPU session A: kernel mix of the 256-graph chem masking step, the bio masking step and the context-prediction step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- "$@" > $O/$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
  rm -rf $O/prof_$name
  python $R/tools/kstats.py $O/${name}_kernel_stats.csv 40 > $O/${name}_kstats.txt
}
prof step_b256 python $R/tools/step_profile.py 256 30 5 epoch
prof bio_step python $R/tools/bio_step_profile.py 256 30
prof ctx_step python $R/tools/ctx_step_profile.py 256 30
cd $R
python tools/step_timeline.py $O/step_b256_trace.csv > $O/step_b256_timeline.txt 2>&1
python tools/step_timeline.py $O/ctx_step_trace.csv > $O/ctx_step_timeline.txt 2>&1
python tools/step_timeline.py $O/bio_step_trace.csv > $O/bio_step_timeline.txt 2>&1
gzip -f $O/*_trace.csv
cat $O/step_b256_kstats.txt
tail -2 $O/step_b256.log $O/bio_step.log $O/ctx_step.log
