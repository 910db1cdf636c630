#!/bin/bash
# round-4 GPU session B: two fp16 planes as the stacks' default (row maxima of hid / dhid from the producing product's epilogue,
# faster forward split): op-level accuracy tests, full suite, step profile
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
rm -f gpurun_out/two_plane_accuracy.jsonl gpurun_out/parity_metrics.jsonl
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "two_fp16 or two_plane or split_weights_2p" > $O/tests_2p_ops.txt 2>&1
tail -n 25 $O/tests_2p_ops.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_ops.py::test_products_on_two_fp16_planes_against_float64 > $O/tests_all.txt 2>&1
tail -n 15 $O/tests_all.txt
cp gpurun_out/two_plane_accuracy.jsonl gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
name=step_b256
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
rm -rf $O/prof_$name
cd $R
python tools/kstats.py $O/${name}_kernel_stats.csv 24 > $O/${name}_kstats.txt
python tools/step_timeline.py $O/${name}_trace.csv > $O/${name}_timeline.txt 2>&1
gzip -f $O/*_trace.csv
cat $O/${name}_kstats.txt
for k in 1 0 1; do
  echo "PGNN_GEMM_2P=$k" >> $O/ab.txt
  PGNN_GEMM_2P=$k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
cat $O/ab.txt
