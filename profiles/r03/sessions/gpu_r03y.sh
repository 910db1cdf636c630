#!/bin/bash
# round-3 GPU session Y: a second sample of the bench line at the final code, and bench.py's multi-rank flow (2 ranks on the one GPU, gloo)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03y
mkdir -p $O
cd $R
PGNN_BENCH_WATCHDOG=60 PGNN_DP_BACKEND=gloo timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_2rank.json 2> $O/bench_2rank.err
echo "2-rank rc=$? bytes=$(wc -c < $O/bench_2rank.json)"
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, b.get("hipgraph_replay",{}).get("ms_per_step"), b.get("contextpred",{}).get("ms_per_step"), b.get("bio_masking",{}).get("ms_per_step"), b.get("bio_masking",{}).get("roofline",{}).get("frac"), b["roofline"]["frac"], b.get("roofline_mlp",{}).get("frac"), b.get("resident_loader",{}).get("ms_per_step"))
c=json.loads(open("$O/bench_2rank.json").read().strip().splitlines()[-1])
print({k:c[k] for k in ("value","ms_per_step","n_gpus")}, c["comm"].get("ms_per_step_by_rank"))
PY
