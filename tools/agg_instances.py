"""HIP-event timing of EVERY chem aggregation instance a 5-layer train step launches, on the roofline batch (bench.py's own
`_time_aggregation`): plain (forward, layer 0), bn_on_read (forward, layers 1-4), transposed (backward, layer 0), transposed_tail
(backward, layers 1-4: + z, BatchNorm-backward sums).  Also the target of the rocprofv3 --pmc passes behind profiles/rNN/agg_pmc_traffic*.json.
usage: [ORDER=survey|smiles] [RELABEL=0|1] python tools/agg_instances.py [graphs=16384]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pretrain_gnns_amd.data import synthetic

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
order = os.environ.get("ORDER", "survey")
big, info = synthetic.chem_aggregation_batch(graphs, order, os.environ.get("RELABEL", "0") != "0", device=dev)
print("batch:", info)
for rep in range(2):
    for which, count, kernel in bench.AGG_INSTANCES:
        ms, per, iters, n, e, alg = bench._time_aggregation(dev, big, which)
        print("%-16s nodes %d edges %d algorithmic bytes %d : %.1f us (min %.1f max %.1f)  %.0f GB/s = %.3f of 8 TB/s" % (
            which, n, e, int(alg), ms * 1e3, float(per.min()) * 1e3, float(per.max()) * 1e3, alg / ms / 1e6, alg / ms / 8e9))
