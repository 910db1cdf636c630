"""host-side (Python) time of each phase of the masking step vs the GPU time of the same phase."""
import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.chem import model as hmodel
from pretrain_gnns_amd.data import synthetic
dev = "cuda"
batch = synthetic.chem_masking_batch(256, seed=7).to(dev)
torch.manual_seed(0)
mods = [hmodel.GNN(5, 300, gnn_type=(sys.argv[1] if len(sys.argv) > 1 else "gin")).to(dev), torch.nn.Linear(300, 119).to(dev), torch.nn.Linear(300, 4).to(dev)]
opts = [torch.optim.Adam(m.parameters(), lr=1e-3, fused=True) for m in mods]
for _ in range(10):
    steps.chem_masking_step(mods, opts, batch)
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
names = ["fwd", "head+loss", "zero", "bwd", "adam", "readback"]
host = [0.0] * 6
gpu = [0.0] * 6
N = 50
for it in range(N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    t = [0.0] * 7
    model, lpa, lpb = mods
    t[0] = time.perf_counter(); ev[0].record()
    node_rep = model(batch.x, batch.edge_index, batch.edge_attr)
    t[1] = time.perf_counter(); ev[1].record()
    pred = lpa(node_rep[batch.masked_atom_indices])
    loss = F.cross_entropy(pred.double(), batch.mask_node_label[:, 0])
    acc = steps._correct(pred, batch.mask_node_label[:, 0])
    t[2] = time.perf_counter(); ev[2].record()
    for o in opts: o.zero_grad()
    t[3] = time.perf_counter(); ev[3].record()
    loss.backward()
    t[4] = time.perf_counter(); ev[4].record()
    for o in opts: o.step()
    t[5] = time.perf_counter(); ev[5].record()
    vals = torch.stack([loss.detach(), acc.double()]).cpu().tolist()
    t[6] = time.perf_counter(); ev[6].record()
    torch.cuda.synchronize()
    for i in range(6):
        host[i] += (t[i + 1] - t[i]) * 1e3
        gpu[i] += ev[i].elapsed_time(ev[i + 1])
for i in range(6):
    print("%-10s host %.3f ms   gpu-span %.3f ms" % (names[i], host[i] / N, gpu[i] / N))
print("total host %.3f gpu %.3f" % (sum(host) / N, sum(gpu) / N))

# ---- how much of the host time is inside the C calls (kernel launches) vs Python/autograd around them
from pretrain_gnns_amd import _lib
lib = _lib.load()
acc = {}
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); acc[self.name] = acc.get(self.name, 0.0) + time.perf_counter() - t0
        return r
class Proxy:
    def __getattr__(self, name):
        return Timed(name, getattr(lib, name))
from pretrain_gnns_amd import ops
ops.load = lambda: Proxy()
for it in range(N):
    steps.chem_masking_step(mods, opts, batch)
torch.cuda.synchronize()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  C call %-34s %.3f ms/step" % (k, v / N * 1e3))
print("  C calls total %.3f ms/step" % (sum(acc.values()) / N * 1e3))
