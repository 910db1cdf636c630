"""exactly `steps` eager context-prediction train steps (256 molecules, device-side ExtractSubstructureContextPair(5,4,7) in the loop)
and nothing else, for `rocprofv3 --kernel-trace --stats`; prints ms/step.
usage: python tools/ctx_step_profile.py [graphs=256] [steps=30]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.chem import model as hmodel
from pretrain_gnns_amd.data import resident, synthetic
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
rng = np.random.default_rng(4321)
graphs = [synthetic.zinc_like_graph(rng) for _ in range(2048)]
ds = resident.ResidentDataset.from_graphs(graphs, dev)
loader = resident.ResidentLoader(ds, g, shuffle=True, seed=2, drop_last=True, substruct_context=(5, 4, 7))
torch.manual_seed(0)
ms_, mc_ = hmodel.GNN(5, 300, gnn_type="gin").to(dev), hmodel.GNN(3, 300, gnn_type="gin").to(dev)
os_, oc_ = bench.make_optimizers((ms_, mc_))
ms_.train(), mc_.train()
it = iter(loader)
for _ in range(3):
    steps.chem_contextpred_step(ms_, mc_, os_, oc_, next(it))
torch.cuda.synchronize()
t0 = time.perf_counter()
done = 0
while done < n_steps:
    for batch in loader:
        steps.chem_contextpred_step(ms_, mc_, os_, oc_, batch)
        done += 1
        if done >= n_steps:
            break
torch.cuda.synchronize()
print("contextpred graphs %d: %.3f ms/step" % (g, (time.perf_counter() - t0) / n_steps * 1e3))
