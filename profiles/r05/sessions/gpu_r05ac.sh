#!/bin/bash
# round-5 GPU session AC: bio -- the BatchNorm backward's sums from the dhid product's epilogue, fork via launch
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ac
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference.py -m gpu -q -x -k "bio" > $O/tests.txt 2>&1
tail -n 5 $O/tests.txt
for rep in 1 2 3; do
for m in 1 0; do
echo "bio PGNN_BN_BWD_IN_GEMM=$m" >> $O/ab.txt
PGNN_BN_BWD_IN_GEMM=$m timeout 300 python tools/bio_step_profile.py 256 60 2>/dev/null | tail -n 1 >> $O/ab.txt
done
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
name=bio_step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python $R/tools/bio_step_profile.py 256 33 > $O/$name.log 2>&1
cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
cp $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) $O/${name}_trace.csv
gzip -f $O/${name}_trace.csv
rm -rf $O/prof_$name
python $R/tools/kstats.py $O/${name}_kernel_stats.csv 22
