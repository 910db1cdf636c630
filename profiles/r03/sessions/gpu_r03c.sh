#!/bin/bash
# round-3 GPU session C: segment-sum / Adam launch changes -- tests of the touched ops, step profile, host-side profile, bench
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q > $O/tests.txt 2>&1
tail -n 5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
name=step_b256
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
rm -rf $O/prof_$name
cd $R
python tools/kstats.py $O/${name}_kernel_stats.csv 45 > $O/${name}_kstats.txt
python tools/step_timeline.py $O/${name}_trace.csv > $O/${name}_timeline.txt 2>&1
python tools/trace_gaps.py $O/${name}_trace.csv > $O/${name}_gaps.txt 2>&1
gzip -f $O/*_trace.csv
timeout 300 python tools/host_profile.py 300 > $O/host_profile.txt 2>&1
head -n 60 $O/host_profile.txt
python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, b.get("hipgraph_replay",{}).get("ms_per_step"), b.get("contextpred",{}).get("ms_per_step"), b.get("bio_masking",{}).get("ms_per_step"))
PY
