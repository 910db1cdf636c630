"""where the HOST time of a context-prediction train step goes (train.chem_contextpred_step, sums on the device, loader in the loop):
ms per step of host enqueue against the step itself, then cProfile top functions.  usage: python tools/ctx_host_profile.py [steps=200]"""
import cProfile, io, os, pstats, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.chem import model as hmodel
from pretrain_gnns_amd.data import resident, synthetic
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
rng = np.random.default_rng(4321)
graphs = [synthetic.zinc_like_graph(rng) for _ in range(2048)]
ds = resident.ResidentDataset.from_graphs(graphs, dev)
loader = resident.ResidentLoader(ds, 256, shuffle=True, seed=2, drop_last=True, substruct_context=(5, 4, 7))
torch.manual_seed(0)
ms_, mc_ = hmodel.GNN(5, 300, gnn_type="gin").to(dev), hmodel.GNN(3, 300, gnn_type="gin").to(dev)
os_, oc_ = bench.make_optimizers((ms_, mc_))
ms_.train(), mc_.train()
accum = steps.epoch_accumulator(dev)


def run(n):
    done = 0
    while done < n:
        for batch in loader:
            steps.chem_contextpred_step(ms_, mc_, os_, oc_, batch, readback="epoch", accum=accum)
            done += 1
            if done >= n:
                break


run(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(n_steps)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("step %.3f ms, host enqueue %.3f ms per step" % ((t2 - t0) / n_steps * 1e3, (t1 - t0) / n_steps * 1e3))
pr = cProfile.Profile()
pr.enable()
run(n_steps)
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(40)
    lines = s.getvalue().splitlines()
    print("\n".join(l[:170] for l in lines[4:54]))
