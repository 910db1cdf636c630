#!/bin/bash
# round-3 GPU session K: HBM traffic (PMC, one pass per counter) of the two aggregation kernels on bench.py's roofline batches
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/a_$c -o agg -- python $R/tools/agg_bench.py 16384 > $O/agg_run_$c.txt 2>&1
  cp $(find $O/a_$c -name "*counter_collection.csv" | head -1) $O/agg_pmc_$c.csv
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/b_$c -o bio -- python $R/tools/bio_tile_pmc.py 40 > $O/bio_run_$c.txt 2>&1
  cp $(find $O/b_$c -name "*counter_collection.csv" | head -1) $O/bio_agg_pmc_$c.csv
  rm -rf $O/a_$c $O/b_$c
done
cd $R
grep -h "nodes\|GB/s" $O/agg_run_FETCH_SIZE.txt | tail -n 4
grep -h "^nodes" $O/bio_run_FETCH_SIZE.txt
python tools/pmc_summary.py "$O/agg_pmc_*.csv" aggregate_dma
python tools/pmc_summary.py "$O/bio_agg_pmc_*.csv" neighbor_sum_tile
