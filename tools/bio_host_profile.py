"""where the HOST time of a bio masking train step goes: cProfile over `steps` eager steps with the device-side loader in the loop
(the bench's bio leg), top functions by cumulative and by own time.  The GPU is not waited for inside the profiled region.
usage: python tools/bio_host_profile.py [graphs=256] [steps=60]"""
import cProfile, io, os, pstats, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.bio import model as hbio
from pretrain_gnns_amd.data import resident, synthetic
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
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
accum = steps.epoch_accumulator(dev)


def run(n):
    done = 0
    while done < n:
        for b in loader:
            steps.bio_masking_step(mods, opts, b, readback="epoch", accum=accum)
            done += 1
            if done >= n:
                break


run(8)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(n_steps)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("loader in the loop: %.3f ms/step, host enqueue %.3f ms/step" % ((t2 - t0) / n_steps * 1e3, (t1 - t0) / n_steps * 1e3))
pr = cProfile.Profile()
pr.enable()
run(n_steps)
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32)
    lines = s.getvalue().splitlines()
    print("\n".join(l[:170] for l in lines[4:48]))
