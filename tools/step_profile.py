"""exactly `steps` eager 256-graph masking train steps and nothing else, for `rocprofv3 --kernel-trace --stats` (the
per-step kernel mix of profiles/rNN/step_b256_kernel_stats.csv = totals / steps).
usage: python tools/step_profile.py [graphs=256] [steps=30] [warmup=5] [readback=epoch]   (prints ms/step measured without the profiler's help)"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.data import synthetic
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
readback = sys.argv[4] if len(sys.argv) > 4 else "epoch"
dev = torch.device("cuda", 0)
mods = bench.make_models(dev)
opts = bench.make_optimizers(mods)
batch = synthetic.chem_masking_batch(graphs, seed=0).to(dev)
step, finish = bench.masking_stepper(mods, opts, readback, dev)
for _ in range(warm):
    step(batch)
finish()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_steps):
    step(batch)
finish()
torch.cuda.synchronize()
print("graphs %d, readback %s: %d steps (+%d warm-up = %d launches of every per-step kernel), %.3f ms/step" % (
    graphs, readback, n_steps, warm, n_steps + warm, (time.perf_counter() - t0) / n_steps * 1e3))
