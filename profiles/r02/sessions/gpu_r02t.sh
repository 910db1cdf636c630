#!/bin/bash
# round-2 GPU session T: HBM traffic of the bio tile aggregation kernel (FETCH_SIZE / WRITE_SIZE, one pass per counter)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$c -o bio -- python $R/tools/bio_tile_pmc.py 40 > $O/run_$c.txt 2>&1
  python $R/tools/pmc_summary.py "$O/p_$c/**/*counter_collection.csv" neighbor_sum_tile > $O/bio_pmc_$c.txt
  cp $(find $O/p_$c -name "*counter_collection.csv" | head -1) $O/bio_agg_pmc_$c.csv
  rm -rf $O/p_$c
  cat $O/bio_pmc_$c.txt; grep "^nodes" $O/run_$c.txt
done
