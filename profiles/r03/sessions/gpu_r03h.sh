#!/bin/bash
# round-3 GPU session H: both tile kernels under test, roofline-only launches (profiled), bench
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_reference.py -m gpu -x -q -k "tile or bio" > $O/tests.txt 2>&1
tail -n 5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
name=roofline_only
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/bench.py --roofline-only > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
rm -rf $O/prof_$name
python $R/tools/kstats.py $O/${name}_kernel_stats.csv 12
tail -n 1 $O/$name.log > $O/roofline_only.json
python - <<PY
import json
d=json.loads(open("$O/roofline_only.json").read())
print({k:(v.get("frac"),v.get("ms_per_launch")) for k,v in d.items() if isinstance(v,dict) and "frac" in v})
PY
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, b.get("hipgraph_replay",{}).get("ms_per_step"), b.get("contextpred",{}).get("ms_per_step"), b.get("bio_masking",{}).get("ms_per_step"), b.get("bio_masking",{}).get("roofline",{}).get("frac"))
PY
