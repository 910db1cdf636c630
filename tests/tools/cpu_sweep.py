import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import chem as ochem, steps
from pretrain_gnns_amd.data import synthetic
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cgroup", e)
batch = synthetic.chem_masking_batch(256, seed=0)
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    torch.manual_seed(0)
    mods = [ochem.GNN(5, 300), torch.nn.Linear(300, 119), torch.nn.Linear(300, 4)]
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in mods]
    steps.chem_masking_step(mods, opts, batch)
    t=time.perf_counter(); n=0
    while time.perf_counter()-t < 4: steps.chem_masking_step(mods, opts, batch); n+=1
    print(th, "threads", (time.perf_counter()-t)/n*1e3, "ms/step", flush=True)
