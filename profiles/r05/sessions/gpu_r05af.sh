#!/bin/bash
# round-5 GPU session AF: PMC passes (one counter set per run, --kernel-trace only): HBM traffic of both aggregation kernels on the
# bench's roofline batches, SQ counters of the 256-graph step's products (the weight-gradient pair has a new instance this round)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05af
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/a_$c -o agg -- python $R/tools/agg_bench.py 16384 > $O/agg_run_$c.txt 2>&1
  cp $(find $O/a_$c -name "*counter_collection.csv" | head -1) $O/agg_pmc_$c.csv
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/b_$c -o bio -- python $R/tools/bio_tile_pmc.py 40 > $O/bio_run_$c.txt 2>&1
  cp $(find $O/b_$c -name "*counter_collection.csv" | head -1) $O/bio_agg_pmc_$c.csv
  rm -rf $O/a_$c $O/b_$c
done
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$tag -o g -- python $R/tools/step_profile.py 256 12 3 epoch > $O/run_$tag.txt 2>&1
  cp $(find $O/p_$tag -name "*counter_collection.csv" | head -1) $O/pmc_step_$tag.csv
  rm -rf $O/p_$tag
done
cd $R
python tools/pmc_summary.py "gpurun_out/r05af/pmc_step_*.csv" "gemm" > $O/pmc_step_summary.txt 2>&1
read N E A <<< $(grep "^nodes" $O/agg_run_FETCH_SIZE.txt | awk '{print $2, $4, $7}')
python tools/pmc_traffic_json.py $O/agg_pmc_FETCH_SIZE.csv $O/agg_pmc_WRITE_SIZE.csv aggregate_dma $N $E $A "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python tools/agg_bench.py 16384 (one pass per counter; tools/gpu_r05af.sh; SMILES-order batch through the loader's renumbering)" > $O/agg_pmc_traffic.json
read N E A <<< $(grep "^nodes" $O/bio_run_FETCH_SIZE.txt | awk '{print $2, $4, $7}')
python tools/pmc_traffic_json.py $O/bio_agg_pmc_FETCH_SIZE.csv $O/bio_agg_pmc_WRITE_SIZE.csv neighbor_sum_tile $N $E $A "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python tools/bio_tile_pmc.py 40 (one pass per counter; tools/gpu_r05af.sh)" > $O/bio_agg_pmc_traffic.json
cat $O/agg_pmc_traffic.json $O/bio_agg_pmc_traffic.json
grep -A13 "k_gemm3_pair" $O/pmc_step_summary.txt | head -16
ls -la $O/*.csv | awk '{print $5, $9}'
rm -f $O/pmc_step_*.csv
gzip -f $O/*_pmc_*.csv
