#!/bin/bash
# round-2 GPU session S: PMC passes over the GEMM kernels at HEAD (MFMA busy / waits; LDS conflicts), separate runs per counter set
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -o gemm -- python $R/tools/gemm_bench.py 6747 262144 > $O/gemm_bench_1.txt 2>&1
python $R/tools/pmc_summary.py "$O/p1/**/*counter_collection.csv" gemm > $O/gemm_pmc_mfma.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/p2 -o gemm -- python $R/tools/gemm_bench.py 6747 262144 > $O/gemm_bench_2.txt 2>&1
python $R/tools/pmc_summary.py "$O/p2/**/*counter_collection.csv" gemm > $O/gemm_pmc_lds.txt
rm -rf $O/p1 $O/p2
cat $O/gemm_pmc_mfma.txt | head -60; cat $O/gemm_pmc_lds.txt | head -40; grep -v rocprof $O/gemm_bench_1.txt | tail -6
