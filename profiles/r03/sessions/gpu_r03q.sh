#!/bin/bash
# round-3 GPU session Q: bench.py's multi-rank flow, functionally, 2 ranks on the one GPU (gloo), several times, with a watchdog that
# dumps the Python stacks if a run exceeds 60 s (one earlier run of this command did not come back)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03q
mkdir -p $O
cd $R
for i in 1 2 3; do
  PGNN_BENCH_WATCHDOG=60 PGNN_DP_BACKEND=gloo timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29520 + i)) bench.py --gpus 2 --steps 10 --warmup 3 > $O/run$i.json 2> $O/run$i.err
  echo "run $i rc=$? bytes=$(wc -c < $O/run$i.json)"
  grep -n "Timeout\|File \"/root\|File \"/tmp\|Thread 0x" $O/run$i.err | head -40
done
