#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --sweep-graphs "" --no-hipgraph --no-loader > $O/bench.json 2>$O/bench.err
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail; cat $O/bench.json | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('step',r['ms_per_step'])
print('ctx',r['contextpred']['ms_per_step'],'bio',r['bio_masking']['ms_per_step'], r['bio_masking']['edges_per_s'], r['bio_masking']['roofline'])"
