"""how long the HOST needs to enqueue a chem masking train step when the device queue is EMPTY at the start (so that a full queue never
blocks the launches): sync, enqueue `burst` steps, stop the clock, sync.  If enqueue/step is well below the step time the device is
the bottleneck and the host runs ahead; if they are equal the step is host-bound.  usage: python tools/host_enqueue.py [graphs=256] [burst=6]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pretrain_gnns_amd.data import synthetic
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
burst = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
mods = bench.make_models(dev)
opts = bench.make_optimizers(mods)
batch = synthetic.chem_masking_batch(graphs, seed=0, device=dev)
step, finish = bench.masking_stepper(mods, opts, "epoch", dev)
for _ in range(200):
    step(batch)
finish()
torch.cuda.synchronize()
enq, tot = [], []
for rep in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(burst):
        step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) / burst * 1e3)
    tot.append((t2 - t0) / burst * 1e3)
enq.sort(); tot.sort()
print("graphs %d burst %d: host enqueue %.3f ms/step (median; min %.3f), burst wall %.3f ms/step (median)" % (graphs, burst, enq[15], enq[0], tot[15]))
t0 = time.perf_counter()
for _ in range(300):
    step(batch)
finish(); torch.cuda.synchronize()
print("steady state %.3f ms/step" % ((time.perf_counter() - t0) / 300 * 1e3))
