"""bench.py's aggregation_robustness leg (same launch, same formula, 16 384 graphs, five atom orders) with the far-row prefetch of
k_aggregate_dma off and on (PGNN_DMA_PF).   usage: python tools/agg_robust_ab.py [graphs=16384]"""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pretrain_gnns_amd import ops
dev = torch.device("cuda", 0)
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for pf in ("0", "1"):
    os.environ["PGNN_DMA_PF"] = pf
    ops.load().pgnn_reload_env()
    r = bench.aggregation_robustness(dev, graphs)
    print("PGNN_DMA_PF=%s " % pf + "  ".join("%s %.1f%% -> %.3f (%.1f us)" % (k, 100 * v["out_of_window_edge_fraction"], v["frac"], 1e3 * v["ms_per_launch"])
                                              for k, v in r.items()), flush=True)
