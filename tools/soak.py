"""soak: many train steps over variable-shape batches from the resident loader; watches loss, memory, status."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops, optim, train as steps
from pretrain_gnns_amd.chem import model as hmodel
from pretrain_gnns_amd.data import resident, synthetic
dev = "cuda"
rng = np.random.default_rng(0)
graphs = [synthetic.zinc_like_graph(rng) for _ in range(8192)]
ds = resident.ResidentDataset.from_graphs(graphs, dev)
loader = resident.ResidentLoader(ds, 256, shuffle=True, seed=1, mask_rate=0.15, drop_last=True)
torch.manual_seed(0)
mods = [hmodel.GNN(5, 300).to(dev), torch.nn.Linear(300, 119).to(dev), torch.nn.Linear(300, 4).to(dev)]
ops.set_direct_grads(True)
opts = optim.Adam.shared([m.parameters() for m in mods], lr=1e-3)  # what bench.py builds
t0 = time.perf_counter(); n = 0; edges = 0
for epoch in range(12):
    if epoch == 1:  # epoch 0 carries module loads and allocator warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0; edges = 0
    out = steps.chem_masking_epoch(mods, opts, loader)
    n += len(loader); edges += sum(int(ds._edges[i].sum()) for i in loader.batch_ids(epoch))
    torch.cuda.synchronize()
    print("epoch %2d loss %.4f acc %.4f  alloc %.1f MB reserved %.1f MB" % (epoch, out[0], out[1], torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20), flush=True)
dt = time.perf_counter() - t0
print("%d steps, %.3f ms/step incl. device-side loader, %.2f M edges/s" % (n, dt / n * 1e3, edges / dt / 1e6))
