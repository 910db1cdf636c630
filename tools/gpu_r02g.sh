#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail; cat $O/bench.json | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('step',r['ms_per_step'],'graph',r['hipgraph_replay']['ms_per_step'],'fwd',r['forward_only']['ms_per_pass'],'loader',r['resident_loader']['ms_per_step'],'ref_loop',r['reference_loop']['ms_per_step'])
print('large',r['large_batch'])
print('ctx',r['contextpred']['ms_per_step'],'bio',r['bio_masking']['ms_per_step'], r['bio_masking']['edges_per_s'])
print('roofline',r['roofline']['frac'],r['roofline_mlp']['achieved'])"
