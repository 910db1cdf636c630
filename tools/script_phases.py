"""where an unchanged chem/pretrain_masking.py step spends its wall clock: the statements of train() (:47-78) around the drop-in GNN,
perf_counter at every statement boundary, averaged; plus the GPU-side duration of the forward and backward alone (events).
usage: python tools/script_phases.py [graphs=256] [steps=200]"""
import os, sys, time, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pretrain_gnns_amd import ops
from pretrain_gnns_amd.data import synthetic

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
batch = synthetic.chem_masking_batch(graphs, seed=0).to(dev)
model, head, bonds = bench.make_models(dev)
opts = [torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0) for m in (model, head, bonds)]
crit = torch.nn.CrossEntropyLoss()
names = ["forward", "head+loss", "accuracy(.item)", "zero_grad", "backward", "adam x3", "loss.item"]
acc = [0.0] * len(names)
model.train()


def step(rec):
    t = [time.perf_counter()]
    node_rep = model(batch.x, batch.edge_index, batch.edge_attr); t.append(time.perf_counter())
    pred = head(node_rep[batch.masked_atom_indices])
    loss = crit(pred.double(), batch.mask_node_label[:, 0]); t.append(time.perf_counter())
    a = float(torch.sum(torch.max(pred.detach(), dim=1)[1] == batch.mask_node_label[:, 0]).cpu().item()) / len(pred); t.append(time.perf_counter())
    for o in opts:
        o.zero_grad()
    t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    for o in opts:
        o.step()
    t.append(time.perf_counter())
    l = float(loss.cpu().item()); t.append(time.perf_counter())
    if rec:
        for i in range(len(names)):
            acc[i] += t[i + 1] - t[i]


for tag, direct in (("as_imported", False), ("PGNN_DIRECT_GRADS=1", True)):
    prev = ops.set_direct_grads(direct)
    for i in range(len(acc)):
        acc[i] = 0.0
    for _ in range(20):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"mode": tag, "ms_per_step": round(total, 4), "phases_ms": {n: round(v / steps * 1e3, 4) for n, v in zip(names, acc)}}), flush=True)
    ops.set_direct_grads(prev)
