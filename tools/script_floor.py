"""how much of the eager step is the script's own torch code?  Same train step with the GNN replaced by a
single embedding lookup (one autograd node, one parameter)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.chem import model as hmodel
from pretrain_gnns_amd.data import synthetic
dev = "cuda"
batch = synthetic.chem_masking_batch(256, seed=7).to(dev)

class Null(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.e = torch.nn.Embedding(120, 300)
    def forward(self, x, ei, ea):
        return self.e(x[:, 0])

for name, gnn in (("null model", Null()), ("5-layer GIN", hmodel.GNN(5, 300))):
    torch.manual_seed(0)
    mods = [gnn.to(dev), torch.nn.Linear(300, 119).to(dev), torch.nn.Linear(300, 4).to(dev)]
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3, fused=True) for m in mods]
    for _ in range(10): steps.chem_masking_step(mods, opts, batch)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 100
    for _ in range(n): steps.chem_masking_step(mods, opts, batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-12s %.3f ms/step" % (name, dt * 1e3))
