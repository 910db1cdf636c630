#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
for c in -1 2; do
PGNN_GEMM3_CFG=$c timeout 300 python tools/gemm_split_check.py 65536 262144 2>$O/err.txt | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l)
    print('cfg $c',r['M'],r['K'],r['N'],' | '.join('%s %s: %s'%(m,k,r[m][k]['us']) for m in ('fp32_mfma', 'split') for k in ('fwd','bwd_data')))"
done
