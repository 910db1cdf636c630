#!/bin/bash
# round-3 GPU session D: edge head + Adam power cache tests, two-workgroups-per-CU tile of k_gemm3w, bio / chem step profiles, bench
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "edge_head or adam or segment or masked_head or embed" > $O/tests_a.txt 2>&1
tail -n 5 $O/tests_a.txt
timeout 900 python -m pytest tests/test_gpu_reference.py tests/test_gpu_models.py -m gpu -x -q -k "bio" > $O/tests_b.txt 2>&1
tail -n 5 $O/tests_b.txt
timeout 300 tools/bin/gemm3w_bench 6747 5100 41269 > $O/gemm3w_bench.txt 2>&1
grep -E "rows|k_gemm3 |cfg" $O/gemm3w_bench.txt | head -40
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- "$@" > $O/$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
  rm -rf $O/prof_$name
  python $R/tools/kstats.py $O/${name}_kernel_stats.csv 45 > $O/${name}_kstats.txt
  python $R/tools/step_timeline.py $O/${name}_trace.csv > $O/${name}_timeline.txt 2>&1
  gzip -f $O/${name}_trace.csv
}
prof step_b256 python $R/tools/step_profile.py 256 30 5 epoch
prof bio_step python $R/tools/bio_step_profile.py 256 30
tail -n 1 $O/bio_step.log
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, b.get("hipgraph_replay",{}).get("ms_per_step"), b.get("contextpred",{}).get("ms_per_step"), b.get("bio_masking",{}).get("ms_per_step"))
PY
