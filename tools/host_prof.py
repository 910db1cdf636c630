"""torch.profiler view of the host side of the masking step (which torch ops / autograd nodes cost CPU time)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.chem import model as hmodel
from pretrain_gnns_amd.data import synthetic
from torch.profiler import profile, ProfilerActivity
dev = "cuda"
batch = synthetic.chem_masking_batch(256, seed=7).to(dev)
torch.manual_seed(0)
mods = [hmodel.GNN(5, 300).to(dev), torch.nn.Linear(300, 119).to(dev), torch.nn.Linear(300, 4).to(dev)]
opts = [torch.optim.Adam(m.parameters(), lr=1e-3, fused=True) for m in mods]
for _ in range(10):
    steps.chem_masking_step(mods, opts, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(20):
        steps.chem_masking_step(mods, opts, batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
