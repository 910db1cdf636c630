#!/bin/bash
# round-4 GPU session AD: bench.py launched as the driver launches it for N = 2 (torch.distributed.run), both ranks on the one GPU over gloo,
# as shipped and with the overlapped all-reduce
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04ad
mkdir -p $O
cd $R
F="--gpus 2 --steps 20 --warmup 5 --no-roofline --no-extra-configs --no-cpu-baseline --no-hipgraph --no-loader --sweep-graphs="
for ov in 0 1; do
  PGNN_DP_BACKEND=gloo PGNN_DP_OVERLAP=$ov timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$ov bench.py $F > $O/bench_n2_ov$ov.json 2> $O/bench_n2_ov$ov.err
  echo "overlap=$ov rc=$?"
  grep "^{" $O/bench_n2_ov$ov.json | tail -n 1 | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print({k:b[k] for k in ('value','n_gpus','ms_per_step')}, b['comm'])"
  tail -n 2 $O/bench_n2_ov$ov.err
done
