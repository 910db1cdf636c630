"""where the HOST time of a chem masking train step goes (train.chem_masking_step, sums on the device): ms per step of host enqueue
against the step itself, then cProfile top functions.  usage: python tools/chem_host_profile.py [steps=300]"""
import cProfile, io, os, pstats, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pretrain_gnns_amd.data import synthetic
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
mods = bench.make_models(dev)
opts = bench.make_optimizers(mods)
batch = synthetic.chem_masking_batch(256, seed=0).to(dev)
step, finish = bench.masking_stepper(mods, opts, "epoch", dev)
for _ in range(20):
    step(batch)
finish()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_steps):
    step(batch)
t1 = time.perf_counter()
finish()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("step %.3f ms, host enqueue %.3f ms per step" % ((t2 - t0) / n_steps * 1e3, (t1 - t0) / n_steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n_steps):
    step(batch)
pr.disable()
finish()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(34)
    lines = s.getvalue().splitlines()
    print("\n".join(l[:170] for l in lines[4:48]))
