#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R; rm -f $O/out.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -p no:cacheprovider -x > $O/pytest.log 2>&1
timeout 300 python tools/gemm_split_check.py 6747 262144 2>$O/err.txt | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l)
    print(r['M'],r['K'],r['N'],' | '.join('%s %s: %s'%(m,k,r[m][k]['us']) for m in ('fp32_mfma', 'split') for k in ('fwd',)), r['split']['fwd'].get('rms_rel_err'))" >> $O/out.txt
timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > $O/bench.json 2>$O/bench.err
tail -5 $O/pytest.log; cat $O/out.txt; tail -3 $O/err.txt; cat $O/bench.json
