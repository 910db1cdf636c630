#!/bin/bash
# round-3 GPU session N (end of round): full GPU suite, smoke, PMC traffic of both aggregation kernels, the profiles behind the bench line, bench
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03n
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/a_$c -o agg -- python $R/tools/agg_bench.py 16384 > $O/agg_run_$c.txt 2>&1
  cp $(find $O/a_$c -name "*counter_collection.csv" | head -1) $O/agg_pmc_$c.csv
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/b_$c -o bio -- python $R/tools/bio_tile_pmc.py 40 > $O/bio_run_$c.txt 2>&1
  cp $(find $O/b_$c -name "*counter_collection.csv" | head -1) $O/bio_agg_pmc_$c.csv
  rm -rf $O/a_$c $O/b_$c
done
prof() {  # name, command...
  local name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- "$@" > $O/$name.log 2>&1
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
timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 > $O/step_unprofiled.txt
timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/step_unprofiled.txt
timeout 100 python tools/ctx_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/step_unprofiled.txt
AB_GRAPHS=8 timeout 100 python tools/host_ab.py 300 2>&1 | tail -n 2 >> $O/step_unprofiled.txt
cat $O/step_unprofiled.txt
python bench.py > $O/bench.json 2> $O/bench.err
python tools/pmc_summary.py "$O/agg_pmc_*.csv" aggregate_dma
python tools/pmc_summary.py "$O/bio_agg_pmc_*.csv" neighbor_sum_tile
head -n 8 $O/roofline_only_kstats.txt
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, b.get("hipgraph_replay",{}).get("ms_per_step"), b.get("contextpred",{}).get("ms_per_step"), b.get("bio_masking",{}).get("ms_per_step"), b.get("bio_masking",{}).get("roofline",{}).get("frac"), b["roofline"]["frac"], b.get("roofline_mlp",{}).get("frac"))
print({k:v["frac"] for k,v in b["aggregation_robustness"].items()})
PY
