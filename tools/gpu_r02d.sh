#!/bin/bash
# round-2 GPU session D: bio tile aggregation tests + A/B of the bio bench leg
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_reference.py -q -p no:cacheprovider -k "tiled or bio" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
for t in 1 0; do
  PGNN_BIO_TILES=$t timeout 600 python - > $O/bio_tiles_$t.json 2> $O/bio_tiles_$t.err <<'PY'
import json, sys, argparse, torch
sys.argv=["bench.py"]
import bench
args = bench.parse()
dev = torch.device("cuda", 0)
print(json.dumps(bench.bio_leg(dev, args, 20, False)))
PY
done
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail; cat $O/bio_tiles_1.json; echo; cat $O/bio_tiles_0.json; tail -3 $O/bio_tiles_1.err
