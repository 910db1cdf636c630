#!/bin/bash
# round-4 GPU session AG: the GPU suite of the final binary
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04ag
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_default.txt 2>&1
tail -n 3 $O/tests_default.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
