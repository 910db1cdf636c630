"""bio (PPI ego-net) masking pre-train step: timing + kernel mix (BASELINE.json configs[4])."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.bio import model as hbio
from pretrain_gnns_amd.data import synthetic
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gnn_type = sys.argv[2] if len(sys.argv) > 2 else "gin"
dev = "cuda"
batch = synthetic.bio_masking_batch(g, seed=0).to(dev)
torch.manual_seed(0)
mods = [hbio.GNN(5, 300, gnn_type=gnn_type).to(dev), torch.nn.Linear(300, 7).to(dev)]
opts = [torch.optim.Adam(m.parameters(), lr=1e-3, fused=True) for m in mods]
for _ in range(5): steps.bio_masking_step(mods, opts, batch)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
for _ in range(n): steps.bio_masking_step(mods, opts, batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("bio %s: graphs %d nodes %d edges %d  %.3f ms/step  %.2f M edges/s" % (gnn_type, g, batch.x.size(0), batch.edge_index.size(1), dt * 1e3, batch.edge_index.size(1) / dt / 1e6))
