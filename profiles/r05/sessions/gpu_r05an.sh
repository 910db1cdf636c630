#!/bin/bash
# round-5 GPU session AN: k_ctx_plan with the BFS in LDS: loader / reference tests (bit-exact against the host extraction), kernel time
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05an
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_loader.py tests/test_gpu_reference.py -m gpu -q > $O/tests.txt 2>&1
tail -n 2 $O/tests.txt
for i in 1 2 3; do timeout 300 python tools/ctx_host_profile.py 200 2>/dev/null | grep "^step"; done | tee $O/ctx.txt
cd /tmp && export TMPDIR=/tmp
name=ctx_step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/ctx_step_profile.py 256 33 > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
rm -rf $O/prof_$name
grep "k_ctx" $O/${name}_kernel_stats.csv | awk -F, '{print $1, $2, $4}' | cut -c1-120
