#!/bin/bash
# round-2 GPU session L: 256-graph step with the split-GEMM tile forced small / big
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in 1 0; do
  echo "== PGNN_GEMM3_CFG=$cfg"
  PGNN_GEMM3_CFG=$cfg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$cfg -o step -- python $R/tools/step_profile.py 256 30 5 > $O/step$cfg.log 2>&1
  python $R/tools/kstats.py $(find $O/prof$cfg -name "*kernel_stats.csv" | head -1) 60 | grep gemm
  rm -rf $O/prof$cfg
  PGNN_GEMM3_CFG=$cfg timeout 100 python $R/tools/step_profile.py 256 200 20 2>&1 | tail -1
done
echo "== default"
timeout 100 python $R/tools/step_profile.py 256 200 20 2>&1 | tail -1
