#!/bin/bash
# round-5 GPU session AL (final binary of the round): the binary as it stands -- whole GPU suite, smoke, the default bench line, rocprofv3 kernel statistics of the
# roofline-only launches, the 256-graph step (+ trace for the timeline), the 16 384-graph step, the bio and context-prediction steps
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05al
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_default.txt 2>&1
tail -n 3 $O/tests_default.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; head -c 400 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
for spec in "roofline_only:bench.py --roofline-only" "step_b256:tools/step_profile.py 256 30 5 epoch" "step_b16384:tools/step_profile.py 16384 8 2 epoch" "bio_step:tools/bio_step_profile.py 256 33" "ctx_step:tools/ctx_step_profile.py 256 33"; do
  name=${spec%%:*}; cmd=${spec#*:}
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/$cmd > $O/$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  if [ "$name" = "step_b256" ]; then cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv; gzip -f $O/${name}_trace.csv; fi
  rm -rf $O/prof_$name
  python $R/tools/kstats.py $O/${name}_kernel_stats.csv 8
  tail -n 1 $O/$name.log | cut -c1-300
done
