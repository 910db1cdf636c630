#!/bin/bash
# round-2 GPU session P: end-of-round artefacts at HEAD -- full bench, rocprof of the roofline launches and of the train step, GEMM checks
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02p
mkdir -p $O
cd $R
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o roofline -- python $R/bench.py --roofline-only > $O/roofline_only.json 2> $O/roofline_only.err
find $O/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/roofline_only_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o step -- python $R/tools/step_profile.py 256 30 5 epoch > $O/step.log 2>&1
find $O/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/step_b256_kernel_stats.csv
python $R/tools/step_timeline.py $(find $O/prof2 -name "*kernel_trace.csv" | head -1) > $O/step_b256_timeline.txt
python $R/tools/trace_gaps.py $(find $O/prof2 -name "*kernel_trace.csv" | head -1) > $O/step_b256_gaps.txt
rm -rf $O/prof1 $O/prof2
cd $R
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -1 > $O/step_unprofiled.txt
timeout 100 python tools/step_profile.py 256 300 20 end 2>&1 | tail -1 >> $O/step_unprofiled.txt
timeout 100 python tools/host_profile.py 300 2>&1 | sed -n 2,2p >> $O/step_unprofiled.txt
timeout 200 python tools/gemm_split_check.py 1000 6747 262144 > $O/gemm_split_check.jsonl 2>/dev/null
timeout 200 python tools/bn_stats_ab.py 256 > $O/bn_stats_ab.txt 2>/dev/null
cut -c1-7000 $O/bench.json; cat $O/step_unprofiled.txt; tail -2 $O/step.log; cut -c1-1200 $O/roofline_only.json
