#!/bin/bash
# round-3 GPU session G: pipelined tile aggregation (loader wave + consumers): bit-identity tests, micro-benchmark, bio roofline
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -k "tile or bio or edge_head or masked_head" > $O/tests.txt 2>&1
tail -n 5 $O/tests.txt
timeout 300 python tools/bio_agg_bench.py > $O/bio_agg_bench.txt 2>&1; cat $O/bio_agg_bench.txt | tail -n 12
PGNN_TILE_PIPE=0 timeout 300 python tools/bio_agg_bench.py > $O/bio_agg_bench_nopipe.txt 2>&1; grep -i "tiled\|neighbour" $O/bio_agg_bench_nopipe.txt | tail -n 6
timeout 300 python tools/bio_step_profile.py 256 30 > $O/bio_step.txt 2>&1; tail -n 1 $O/bio_step.txt
PGNN_TILE_PIPE=0 timeout 300 python tools/bio_step_profile.py 256 30 > $O/bio_step_nopipe.txt 2>&1; tail -n 1 $O/bio_step_nopipe.txt
