#!/bin/bash
# round-4 GPU session AC: the driver's window (K = 20, W = 5) with and without the settle steps in front of the warm-up
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04ac
mkdir -p $O
cd $R
F="--gpus 1 --steps 20 --warmup 5 --no-roofline --no-extra-configs --no-cpu-baseline --no-hipgraph --no-loader --sweep-graphs="
for s in 0 200 0 200 0 200; do
  python bench.py $F --settle-steps $s 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('settle $s: value %.2f M edges/s, %.4f ms/step; windows median %.4f' % (b['value']/1e6, b['ms_per_step'], b['value_windows']['ms_per_step_median']))" | tee -a $O/ab.txt
done
