#!/bin/bash
# round-2 GPU session I: end-of-round artefacts -- full GPU suite, full bench, rocprof of the roofline launches and of the train step
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o roofline -- python $R/bench.py --roofline-only > $O/roofline_only.json 2> $O/roofline_only.err
find $O/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/roofline_only_kernel_stats_v2.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o step -- python $R/tools/step_profile.py 256 30 5 > $O/step.log 2>&1
find $O/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/step_b256_kernel_stats_v2.csv
find $O/prof2 -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/step_b256_kernel_trace_v2.csv
rm -rf $O/prof1 $O/prof2
cd $R
timeout 300 python tools/gemm_split_check.py 1000 6747 262144 > $O/gemm_split_check.jsonl 2>/dev/null
timeout 300 python tools/gemm_ksweep.py > $O/gemm_ksweep.txt 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -5; cat $O/bench.json | cut -c1-6000; tail -2 $O/step.log; cat $O/roofline_only.json | cut -c1-1500
