#!/bin/bash
# round-4 GPU session C: BatchNorm-backward sums from the transposed aggregation's epilogue; relabelled-dataset tests; bench legs on the SMILES-order batch
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_loader.py -m gpu -q -x -k "transposed_aggregation or relabelled or one_call" > $O/tests_new.txt 2>&1
tail -n 12 $O/tests_new.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_ops.py::test_products_on_two_fp16_planes_against_float64 > $O/tests_all.txt 2>&1
tail -n 12 $O/tests_all.txt
for k in 1 0 1 0; do
  echo "PGNN_BN_BWD_IN_AGG=$k" >> $O/ab.txt
  PGNN_BN_BWD_IN_AGG=$k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
name=step_b256
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/step_profile.py 256 30 5 epoch > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
rm -rf $O/prof_$name
cd $R
python tools/kstats.py $O/${name}_kernel_stats.csv 16 > $O/${name}_kstats.txt
python tools/step_timeline.py $O/${name}_trace.csv > $O/${name}_timeline.txt 2>&1
gzip -f $O/*_trace.csv
cat $O/${name}_kstats.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -n 5 $O/bench.err
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")})
r=b.get("roofline",{}); print("roofline", r.get("frac"), r.get("batch"), "as_fed", r.get("as_fed",{}).get("frac"), "survey", r.get("survey_order",{}).get("frac"))
for k,v in b.get("aggregation_robustness",{}).items(): print(k, v.get("out_of_window_edge_fraction"), v.get("frac"))
print("ctx", b.get("contextpred",{}).get("ms_per_step"), "bio", b.get("bio_masking",{}).get("ms_per_step"), "3p", b.get("three_plane_products"))
PY
