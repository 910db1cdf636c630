#!/bin/bash
# round-5 GPU session A: the fused mlp kernel (csrc/mlp_fused.hip) -- parity tests, then time against the two products it replaces
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
timeout 120 python tools/mlp_fused_bench.py 6747 262144 --iters 5 > $O/first.txt 2>&1
tail -n 4 $O/first.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused_mlp" > $O/tests_fused.txt 2>&1
tail -n 15 $O/tests_fused.txt
timeout 300 python tools/mlp_fused_bench.py 262144 131072 65536 32768 16384 8192 6747 --iters 20 --out $O/mlp_fused_ab.jsonl > $O/bench.txt 2>&1
tail -n 8 $O/bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -o 'SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CU_CYCLES\|SQ_WAIT_INST_LDS\|SQ_LDS_BANK_CONFLICT\|SQ_LDS_IDX_ACTIVE\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*\|SQ_ACTIVE_INST_LDS' $O/counters.txt | sort -u > $O/counter_names.txt
cat $O/counter_names.txt
