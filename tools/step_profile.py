"""run a few train steps at a given per-GPU batch (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.chem import model as hmodel
from pretrain_gnns_amd.data import synthetic
g = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = "cuda"
base = synthetic.chem_masking_batch(min(g, 2048), seed=7)
batch = (synthetic.tile_batch(base, g // 2048) if g > 2048 else base).to(dev)
torch.manual_seed(0)
mods = [hmodel.GNN(5, 300).to(dev), torch.nn.Linear(300, 119).to(dev), torch.nn.Linear(300, 4).to(dev)]
opts = [torch.optim.Adam(m.parameters(), lr=1e-3, fused=True) for m in mods]
for _ in range(26):
    steps.chem_masking_step(mods, opts, batch)
torch.cuda.synchronize()
