#!/bin/bash
# round-3 GPU session J: full GPU suite + the profiles behind the bench line (roofline-only launches, the three step kernel mixes) + bench
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- "$@" > $O/$name.log 2>&1
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
grep "^{" $O/roofline_only.log | tail -n 1 > $O/roofline_only.json
python bench.py > $O/bench.json 2> $O/bench.err
head -n 14 $O/roofline_only_kstats.txt
for n in step_b256 bio_step ctx_step; do grep -v "^W\|^E\|^I" $O/$n.log | tail -n 1; done
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, b.get("hipgraph_replay",{}).get("ms_per_step"), b.get("contextpred",{}).get("ms_per_step"), b.get("bio_masking",{}).get("ms_per_step"), b.get("bio_masking",{}).get("roofline",{}).get("frac"), b["roofline"]["frac"], b.get("roofline_mlp",{}).get("frac"))
PY
