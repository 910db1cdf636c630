#!/bin/bash
# round-5 GPU session F: the wide aggregation window (POL bit 5) + relabel default + milestone keyed on the network: tests, roofline legs
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_parallel.py tests/test_gpu_loader.py -m gpu -q -x -k "aggregate or wide or milestone or relabel or loader or neighbor or gcn" > $O/tests_sel.txt 2>&1
tail -n 6 $O/tests_sel.txt
for w in 1 0; do
  echo "PGNN_DMA_WIDE=$w" | tee -a $O/roof.txt
  PGNN_DMA_WIDE=$w timeout 600 python bench.py --roofline-only 2>&1 | tail -n 1 > $O/roof_wide$w.json
  python - <<PY | tee -a $O/roof.txt
import json
j=json.loads(open("$O/roof_wide$w.json").read())
r=j.get("roofline",j)
print("frac",r.get("frac"),"ms",r.get("ms_per_launch"),"as_fed",r.get("as_fed",{}).get("frac"),"survey",r.get("survey_order",{}).get("frac"))
a=j.get("aggregation_robustness")
if a:
    for k,v in a.items():
        if isinstance(v,dict): print(" ",k,v.get("out_of_window_edge_fraction"),v.get("frac"))
PY
done
timeout 300 python tools/script_phases.py 256 200 2>&1 | tail -n 2 | tee $O/script_phases.txt
