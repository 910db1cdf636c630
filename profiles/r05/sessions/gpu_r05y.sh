#!/bin/bash
# round-5 GPU session Y: the substructure/context loader plans batch t + 1 in front of the train step of batch t (no per-step drain)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05y
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_loader.py tests/test_gpu_reference.py -m gpu -q -x > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
python tools/ctx_host_profile.py 200 > $O/ctx_host.txt 2>&1
head -3 $O/ctx_host.txt
for i in 1 2 3; do timeout 300 python tools/ctx_step_profile.py 256 100 2>/dev/null | tail -n 1; done | tee $O/ctx_steps.txt
