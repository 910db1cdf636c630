#!/bin/bash
# round-2 GPU session B: full -m gpu suite again (new parity bars, 2-rank test), bench with the new legs
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
rm -f gpurun_out/parity_metrics.jsonl
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
cp gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -15; tail -3 $O/bench.err; cat $O/bench.json | head -c 3000
