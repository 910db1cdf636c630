#!/bin/bash
# round-4 GPU session S: memory-side counters of k_gemm2pr at 262 144 rows (fabric reads, L2 hit rate, writes)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  PGNN_GEMM2P_RES=2 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$tag -o g -- python $R/tools/gemm2pr_once.py 262144 4 > $O/run_$tag.txt 2>&1
  cp $(find $O/p_$tag -name "*counter_collection.csv" | head -1) $O/pmc_$tag.csv
  rm -rf $O/p_$tag
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r04s/pmc_*.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm2pr" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f.split("/")[-1], k, "per launch", sum(v) / len(v), "launches", len(v))
PY
