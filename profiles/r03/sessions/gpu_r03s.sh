#!/bin/bash
# round-3 GPU session S: the single-workgroup kernels with batched loads -- their tests, A/B of the unprofiled steps, their durations
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03s
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "graph or group_by or edge_head or embed" > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
for k in 1 0 1 0; do
  echo "PGNN_GRAPH_SMALL=$k" >> $O/ab.txt
  PGNN_GRAPH_SMALL=$k timeout 100 python tools/step_profile.py 256 300 20 epoch 2>&1 | tail -n 1 >> $O/ab.txt
done
for k in 1 0; do
  echo "PGNN_GROUP_SMALL=$k" >> $O/ab.txt
  PGNN_GROUP_SMALL=$k timeout 100 python tools/bio_step_profile.py 256 100 2>&1 | tail -n 1 >> $O/ab.txt
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- "$@" > $O/$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  rm -rf $O/prof_$name
  python $R/tools/kstats.py $O/${name}_kernel_stats.csv 60 > $O/${name}_kstats.txt
}
prof step_b256 python $R/tools/step_profile.py 256 30 5 epoch
prof bio_step python $R/tools/bio_step_profile.py 256 30
grep -E "k_chem_graph_small|k_group_small|k_bio_payload" $O/step_b256_kstats.txt $O/bio_step_kstats.txt
