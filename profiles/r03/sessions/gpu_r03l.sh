#!/bin/bash
# round-3 GPU session L: backward on the calling thread, counters inside the stack call: full GPU suite, host A/B (tiny batch), bench
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
AB_GRAPHS=8 timeout 200 python tools/host_ab.py 300 2>&1 | tail -n 4
timeout 200 python tools/host_ab.py 300 2>&1 | tail -n 2
python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, b.get("hipgraph_replay",{}).get("ms_per_step"), b.get("contextpred",{}).get("ms_per_step"), b.get("bio_masking",{}).get("ms_per_step"), b.get("bio_masking",{}).get("roofline",{}).get("frac"), b["roofline"]["frac"], b["roofline"].get("traffic"))
PY
