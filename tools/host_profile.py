"""cProfile view of the host side of an eager 256-graph masking train step (Python / ctypes / torch time per call; the torch.profiler view is tools/host_prof.py)
usage: python tools/host_profile.py [steps=200]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pretrain_gnns_amd import ops, train as steps
from pretrain_gnns_amd.data import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
ops.set_direct_grads(True)
mods = bench.make_models(dev)
opts = bench.make_optimizers(mods)
batch = synthetic.chem_masking_batch(256, seed=0).to(dev)
accum = steps.epoch_accumulator(dev)
for _ in range(20):
    steps.chem_masking_step(mods, opts, batch, readback="epoch", accum=accum)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    steps.chem_masking_step(mods, opts, batch, readback="epoch", accum=accum)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, with final sync %.3f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    steps.chem_masking_step(mods, opts, batch, readback="epoch", accum=accum)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(40)
st.sort_stats("tottime").print_stats(25)
