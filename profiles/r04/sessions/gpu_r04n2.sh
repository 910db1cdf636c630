#!/bin/bash
# round-4 GPU session N2 (evidence, timing part, final binary): the profiles behind the bench line (roofline-only launches, the three step
# kernel mixes), unprofiled steps, bench.  (tests, smoke and the PMC passes: tools/gpu_r04m.sh, same binary, session r04m2 -- a box
# whose kernel-dispatch gaps were ~2.5 us longer: 1.21 ms steps with the same kernel durations; its bench line is kept as bench_v4_slow_box.json)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04n3
mkdir -p $O
cd $R
(grep "model name" /proc/cpuinfo | head -1; nproc) > $O/host.txt
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 | tee -a $O/step_unprofiled.txt
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- "$@" > $O/$name.log 2>&1
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
for k in 1 2; do timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/step_unprofiled.txt; done
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/step_unprofiled.txt
timeout 100 python tools/ctx_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/step_unprofiled.txt
cat $O/step_unprofiled.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -n 3 $O/bench.err
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, "windows", b.get("value_windows",{}).get("ms_per_step_median"))
print("replay", b.get("hipgraph_replay",{}).get("ms_per_step"), "ctx", b.get("contextpred",{}).get("ms_per_step"), "bio", b.get("bio_masking",{}).get("ms_per_step"))
r=b["roofline"]; print("roofline", r["frac"], "as_fed", r.get("as_fed",{}).get("frac"), "survey", r.get("survey_order",{}).get("frac"))
m=b.get("roofline_mlp",{}); print("mlp", m.get("achieved"), m.get("frac"), m.get("ms_per_launch"), "tiled", m.get("tiled_two_plane_kernel",{}).get("ms_per_launch"), "3p", m.get("three_plane_kernel",{}).get("ms_per_launch"))
print("unchanged", b.get("unchanged_script"))
print("large", b.get("large_batch"))
PY
