"""where the time of an UNCHANGED chem/pretrain_masking.py step goes (bench.py's unchanged_script leg): wall-clock sections of the
script's own statements, then cProfile of the host side.  usage: python tools/script_host_profile.py [steps=200] [direct=0|1]"""
import cProfile, io, os, pstats, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pretrain_gnns_amd import ops
from pretrain_gnns_amd.data import synthetic
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
direct = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
dev = torch.device("cuda", 0)
batch = synthetic.chem_masking_batch(256, seed=0).to(dev)
model, lin_atoms, lin_bonds = bench.make_models(dev)
opts = [torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0) for m in (model, lin_atoms, lin_bonds)]
criterion = torch.nn.CrossEntropyLoss()
ops.set_direct_grads(direct)
model.train()
sec = {k: 0.0 for k in ("forward", "head+loss", "accuracy(sync)", "zero_grad", "backward", "adam", "loss.item(sync)")}


def one_step(timed):
    t = [time.perf_counter()]
    node_rep = model(batch.x, batch.edge_index, batch.edge_attr); t.append(time.perf_counter())
    pred_node = lin_atoms(node_rep[batch.masked_atom_indices])
    loss = criterion(pred_node.double(), batch.mask_node_label[:, 0]); t.append(time.perf_counter())
    acc = float(torch.sum(torch.max(pred_node.detach(), dim=1)[1] == batch.mask_node_label[:, 0]).cpu().item()) / len(pred_node); t.append(time.perf_counter())
    for o in opts:
        o.zero_grad()
    t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    for o in opts:
        o.step()
    t.append(time.perf_counter())
    v = float(loss.cpu().item()); t.append(time.perf_counter())
    if timed:
        for k, a, b in zip(sec, t[:-1], t[1:]):
            sec[k] += b - a
    return v, acc


for _ in range(10):
    one_step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_steps):
    one_step(True)
torch.cuda.synchronize()
print("direct_grads=%d: %.3f ms/step" % (direct, (time.perf_counter() - t0) / n_steps * 1e3))
print("  ".join("%s %.3f" % (k, v / n_steps * 1e3) for k, v in sec.items()), "(ms per step, host wall clock of each statement group)")
pr = cProfile.Profile()
pr.enable()
for _ in range(n_steps):
    one_step(False)
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(30)
    lines = s.getvalue().splitlines()
    print("\n".join(l[:160] for l in lines[4:44]))
