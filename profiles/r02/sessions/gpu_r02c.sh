#!/bin/bash
# round-2 GPU session C: -m gpu suite (attention kernels, GCN on the DMA kernel, bio device transform), rocprof evidence
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
rm -f gpurun_out/parity_metrics.jsonl
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
cp gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roof -o roof -- python $R/bench.py --roofline-only > $O/roofline_only.json 2> $O/roofline_only.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o step -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-hipgraph --no-loader --no-extra-configs --sweep-graphs '' > $O/step.json 2> $O/step.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o agg -- python $R/tools/agg_bench.py 16384 > $O/pmc_$c.log 2>&1
done
cd $R
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -15; cat $O/roofline_only.json | head -c 1500; ls $O/prof_roof $O/pmc_FETCH_SIZE 2>/dev/null | head
