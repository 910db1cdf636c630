#!/bin/bash
# round-5 GPU session D: fused mlp with the refill DMA interleaved in the MFMA loops: tests, timing, SQ counters
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused_mlp" > $O/tests_fused.txt 2>&1
tail -n 5 $O/tests_fused.txt
timeout 300 python tools/mlp_fused_bench.py 262144 65536 32768 --iters 20 --out $O/mlp_fused_ab.jsonl > $O/bench.txt 2>&1
tail -n 4 $O/bench.txt
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$tag -o g -- python $R/tools/mlp_fused_bench.py 262144 --iters 3 > $O/run_$tag.txt 2>&1
  cp $(find $O/p_$tag -name "*counter_collection.csv" | head -1) $O/pmc_$tag.csv
  rm -rf $O/p_$tag
done
cd $R
python tools/pmc_summary.py "gpurun_out/r05d/pmc_*.csv" "" > $O/pmc_summary.txt 2>&1
grep -A14 'k_mlp2p_fused\|k_gemm2pr' $O/pmc_summary.txt | head -120
