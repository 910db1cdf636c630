"""host-side A/B of the eager 256-graph masking step: autograd engine on its own thread (default) vs on the calling thread
(torch.autograd.set_multithreading_enabled(False)); wall per step with the GPU kept busy (epoch read-back) and host-only enqueue time
usage: python tools/host_ab.py [steps=300]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pretrain_gnns_amd import ops, train as steps
from pretrain_gnns_amd.data import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
mods = bench.make_models(dev)
opts = bench.make_optimizers(mods)
batch = synthetic.chem_masking_batch(int(os.environ.get("AB_GRAPHS", "256")), seed=0, device=dev)
accum = steps.epoch_accumulator(dev)


def run(k):
    for _ in range(k):
        steps.chem_masking_step(mods, opts, batch, readback="epoch", accum=accum)


def measure(tag):
    run(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-34s host enqueue %.3f ms/step, wall %.3f ms/step" % (tag, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)


import gc
gc.collect(); gc.disable()
for rep in range(2):
    measure("engine thread (default)")
    with torch.autograd.set_multithreading_enabled(False):
        measure("backward on the calling thread")
