#!/bin/bash
# round-4 GPU session H (evidence): full GPU suite, smoke, PMC traffic of the chem aggregation on the SMILES-order roofline batch,
# the profiles behind the bench line (roofline-only launches, the three step kernel mixes), unprofiled steps, bench
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
rm -f gpurun_out/parity_metrics.jsonl gpurun_out/two_plane_accuracy.jsonl
timeout 1200 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
cp gpurun_out/parity_metrics.jsonl gpurun_out/two_plane_accuracy.jsonl $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/a_$c -o agg -- python $R/tools/agg_bench.py 16384 > $O/agg_run_$c.txt 2>&1
  cp $(find $O/a_$c -name "*counter_collection.csv" | head -1) $O/agg_pmc_$c.csv
  rm -rf $O/a_$c
done
prof() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- "$@" > $O/$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
  rm -rf $O/prof_$name
  python $R/tools/kstats.py $O/${name}_kernel_stats.csv 45 > $O/${name}_kstats.txt
}
prof roofline_only python $R/bench.py --roofline-only
prof step_b256 python $R/tools/step_profile.py 256 30 5 epoch
prof bio_step python $R/tools/bio_step_profile.py 256 30
prof ctx_step python $R/tools/ctx_step_profile.py 256 30
cd $R
for n in step_b256 ctx_step bio_step; do python tools/step_timeline.py $O/${n}_trace.csv > $O/${n}_timeline.txt 2>&1; done
python tools/trace_gaps.py $O/step_b256_trace.csv > $O/step_b256_gaps.txt 2>&1
gzip -f $O/*_trace.csv
grep "^{" $O/roofline_only.log | tail -n 1 > $O/roofline_only.json
for k in "PGNN_X=0" "PGNN_BN_APPLY_BPC=4" "PGNN_X=0" "PGNN_BN_APPLY_BPC=4"; do
  echo "$k" >> $O/step_unprofiled.txt
  env $k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/step_unprofiled.txt
done
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/step_unprofiled.txt
timeout 100 python tools/ctx_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/step_unprofiled.txt
cat $O/step_unprofiled.txt
read N E A <<< $(grep "^nodes" $O/agg_run_FETCH_SIZE.txt | awk '{print $2, $4, $7}')
python tools/pmc_traffic_json.py $O/agg_pmc_FETCH_SIZE.csv $O/agg_pmc_WRITE_SIZE.csv aggregate_dma $N $E $A "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python tools/agg_bench.py 16384 (one pass per counter; tools/gpu_r04h.sh; SMILES-order batch through the loader's renumbering)" > $O/agg_pmc_traffic.json
cat $O/agg_pmc_traffic.json | head -20
mkdir -p profiles/r04 && cp $O/agg_pmc_traffic.json profiles/r04/agg_pmc_traffic.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -n 3 $O/bench.err
head -n 10 $O/roofline_only_kstats.txt
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, "windows", b.get("value_windows"))
print("replay", b.get("hipgraph_replay",{}).get("ms_per_step"), "ctx", b.get("contextpred",{}).get("ms_per_step"), "bio", b.get("bio_masking",{}).get("ms_per_step"), "bio roofline", {k:b.get("bio_masking",{}).get("roofline",{}).get(k) for k in ("frac","frac_on_own_bytes","frac_on_traffic")})
r=b["roofline"]; print("roofline", r["frac"], "traffic", r.get("traffic"), r.get("batch"), "as_fed", r.get("as_fed",{}).get("frac"), "survey", r.get("survey_order",{}).get("frac"))
m=b.get("roofline_mlp",{}); print("mlp", m.get("achieved"), m.get("frac"), m.get("frac_of_fp32_mfma_peak"), m.get("ms_per_launch"), "3p", m.get("three_plane_kernel",{}).get("ms_per_launch"))
print("mlp_step", b.get("roofline_mlp_step"))
print("unchanged", b.get("unchanged_script"), "reference_loop", b.get("reference_loop"))
print("large", b.get("large_batch"))
print({k:(v["out_of_window_edge_fraction"], v["frac"]) for k,v in b["aggregation_robustness"].items()})
print("3p", b.get("three_plane_products"), "loader", b.get("resident_loader"), "fwd", b.get("forward_only"))
PY
