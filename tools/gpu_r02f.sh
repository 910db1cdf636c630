#!/bin/bash
# round-2 GPU session F: split-bf16 GEMM accuracy and speed
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
timeout 300 python tools/gemm_split_check.py 6747 1000 > $O/small.jsonl 2> $O/small.err
for cfg in 0 1 2; do for st in 1 2; do
  PGNN_GEMM3_CFG=$cfg PGNN_GEMM3_STAGES=$st timeout 300 python tools/gemm_split_check.py 6747 262144 > $O/cfg${cfg}_st${st}.jsonl 2> $O/cfg${cfg}_st${st}.err
done; done
tail -2 $O/small.err; cat $O/small.jsonl; for f in $O/cfg*.jsonl; do echo $f; python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    r=json.loads(l)
    print(r["M"],r["K"],r["N"]," | ".join("%s %s: %s"%(m,k,r[m][k]["us"]) for m in ("fp32_mfma","split") for k in ("fwd","bwd_data","bwd_weight")))
PY
done
