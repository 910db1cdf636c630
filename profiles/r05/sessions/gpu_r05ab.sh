#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ab
mkdir -p $O
cd $R
run() {
  echo "$1" >> $O/ctx_ab.txt
  env $1 timeout 300 python tools/ctx_host_profile.py 200 2>/dev/null | grep "^step" >> $O/ctx_ab.txt
}
for rep in 1 2 3; do
  run "PGNN_X=0"
  run "PGNN_SIDE_MIN_ROWS=0"
  run "PGNN_SIDE_MIN_ROWS=0 PGNN_STACK_PER_LAYER_BUFFERS=1"
done
cat $O/ctx_ab.txt
python - <<'PY'
import time, os
# host speed probe: python loop + a trivial HIP call rate
import torch
t0=time.perf_counter(); s=0
for i in range(2000000): s+=i
t1=time.perf_counter()
x=torch.zeros(8,device='cuda'); torch.cuda.synchronize()
t2=time.perf_counter()
for i in range(20000): x.add_(1)
t3=time.perf_counter(); torch.cuda.synchronize()
print("host probe: python loop %.1f ns/iter, torch add_ enqueue %.2f us/call"%((t1-t0)/2e6*1e9,(t3-t2)/2e4*1e6))
print(open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0])
PY
