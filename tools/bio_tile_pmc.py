"""the bio GINConv aggregate (graph-resident tile kernel: neighbour sum + edge-feature product in one launch) alone on bench.py's
bio roofline batch (4096 PPI-ego-shaped graphs), for `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE` (one pass per counter)
usage: python tools/bio_tile_pmc.py [launches=40]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops
from pretrain_gnns_amd.data import resident, synthetic
dev = torch.device("cuda", 0)
rng = np.random.default_rng(99)  # the generator state bench.py's bio leg draws its 1024 graphs from
graphs = [synthetic.ppi_like_graph(rng) for _ in range(1024)]
ds = resident.ResidentDataset.from_graphs(graphs, dev)
big = ds.collate(np.arange(4096) % len(graphs))
n, e = big.x.size(0), big.edge_index.size(1)
graph = ops.build_bio_graph(big.edge_index, big.edge_attr, n, gcn=False)
x = torch.randn(n, 300, device=dev)
enc_w, enc_b = torch.randn(300, 9, device=dev), torch.randn(300, device=dev)
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
        ops.BioAggregate.apply(x, enc_w, enc_b, graph)
torch.cuda.synchronize()
print("nodes %d edges %d algorithmic bytes %d" % (n, e, 3604 * n + 40 * e))
