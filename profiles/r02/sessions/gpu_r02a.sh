#!/bin/bash
# round-2 GPU session A: full -m gpu suite (incl. the reference-fixture parity tests), aggregation policy sweep +
# phase instrumentation, default bench line, LDS-conflict PMC pass on the current GEMM
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
rm -f gpurun_out/parity_metrics.jsonl
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
cp gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
timeout 600 python tools/agg_sweep.py 16384 60 > $O/agg_sweep.log 2>&1
cp gpurun_out/agg_sweep.json $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/gemm_pmc_lds -- python $R/tools/gemm_bench.py 262144 > $O/gemm_pmc.log 2>&1
tail -5 $O/pytest.log; tail -20 $O/agg_sweep.log; cat $O/bench.json | head -c 1500
