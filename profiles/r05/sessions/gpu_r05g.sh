#!/bin/bash
# round-5 GPU session G: (1) tightened two-plane / fused tests; (2) SQ counters of the small-M products and the weight-gradient pair;
# (3) compile-time ablations of the fused mlp kernel (an A/B build of mlp_fused.hip made on the box, discarded with it)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "two_fp16_planes or fused_mlp or resident_plane" > $O/tests_planes.txt 2>&1
tail -n 4 $O/tests_planes.txt
timeout 300 python tools/script_phases.py 256 200 2>&1 | tail -n 2 | tee $O/script_phases_default.txt
PGNN_SIDE_STREAM=0 timeout 300 python tools/script_phases.py 256 200 2>&1 | tail -n 2 | tee $O/script_phases_one_stream.txt
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$tag -o g -- python $R/tools/step_profile.py 256 12 3 epoch > $O/run_$tag.txt 2>&1
  cp $(find $O/p_$tag -name "*counter_collection.csv" | head -1) $O/pmc_step_$tag.csv
  rm -rf $O/p_$tag
done
cd $R
python tools/pmc_summary.py "gpurun_out/r05g/pmc_step_*.csv" "gemm" > $O/pmc_step_summary.txt 2>&1
grep -c . $O/pmc_step_summary.txt
cd $R/pretrain_gnns_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DPGNN_AB -c mlp_fused.hip -o _obj/mlp_fused.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpgnn.so _obj/*.o
cd $R
for abl in 0 1 2 3 4 7 8 15 16 31 0; do
  PGNN_FUSED_ABL=$abl timeout 120 python tools/mlp_fused_bench.py 262144 65536 --iters 10 --fused-only --out $O/ablation.jsonl 2>&1 | tail -n 2
done
