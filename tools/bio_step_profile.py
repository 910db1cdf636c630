"""exactly `steps` eager bio masking train steps (256 PPI-ego-shaped graphs, device-side collate + MaskEdge in the loop) and nothing
else, for `rocprofv3 --kernel-trace --stats`; prints ms/step and the host-only enqueue time.
usage: python tools/bio_step_profile.py [graphs=256] [steps=30]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.bio import model as hbio
from pretrain_gnns_amd.data import resident, synthetic
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
rng = np.random.default_rng(99)
graphs = [synthetic.ppi_like_graph(rng) for _ in range(1024)]
ds = resident.ResidentDataset.from_graphs(graphs, dev)
loader = resident.ResidentLoader(ds, g, shuffle=True, seed=3, mask_rate=0.15, drop_last=True)
torch.manual_seed(0)
mods = [hbio.GNN(5, 300, gnn_type="gin").to(dev), torch.nn.Linear(300, 7).to(dev)]
opts = bench.make_optimizers(mods)
for m in mods:
    m.train()
batches = [b for b in loader]
accum = steps.epoch_accumulator(dev)
for b in batches[:3]:
    steps.bio_masking_step(mods, opts, b, readback="epoch", accum=accum)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n_steps):
    steps.bio_masking_step(mods, opts, batches[i % len(batches)], readback="epoch", accum=accum)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("bio graphs %d nodes %d edges %d: %.3f ms/step (host enqueue %.3f ms/step)" % (g, batches[0].x.size(0), batches[0].edge_index.size(1), (t2 - t0) / n_steps * 1e3, (t1 - t0) / n_steps * 1e3))
