"""masking pre-train step time for every gnn_type (chem, 256 graphs) -- informational."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.chem import model as hmodel
from pretrain_gnns_amd.data import synthetic
dev = "cuda"
batch = synthetic.chem_masking_batch(256, seed=7).to(dev)
for gnn_type in ("gin", "gcn", "graphsage", "gat"):
    torch.manual_seed(0)
    mods = [hmodel.GNN(5, 300, gnn_type=gnn_type).to(dev), torch.nn.Linear(300, 119).to(dev), torch.nn.Linear(300, 4).to(dev)]
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3, fused=True) for m in mods]
    for _ in range(5): steps.chem_masking_step(mods, opts, batch)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 30
    for _ in range(n): steps.chem_masking_step(mods, opts, batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("chem %-9s %.3f ms/step  %.2f M edges/s" % (gnn_type, dt * 1e3, batch.edge_index.size(1) / dt / 1e6))
