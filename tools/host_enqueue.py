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

if os.environ.get("LOADER", "1") != "0":  # the same measurement with the resident loader handing out a NEW batch every step
    import numpy as np
    from pretrain_gnns_amd.data import resident
    rng = np.random.default_rng(1234)
    ds = resident.ResidentDataset.from_graphs([synthetic.zinc_like_graph(rng) for _ in range(4096)], dev)
    loader = resident.ResidentLoader(ds, graphs, shuffle=True, seed=1, mask_rate=0.15, drop_last=True)
    for _ in range(3):
        for b in loader:
            step(b)
    finish(); torch.cuda.synchronize()
    enq, tot = [], []
    for rep in range(12):
        it = iter(loader)
        first = next(it)  # (the epoch's upload and the first, un-overlapped collate stay outside the burst)
        step(first)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        for b in it:
            step(b)
            k += 1
            if k == burst:
                break
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for b in it:
            pass
        enq.append((t1 - t0) / burst * 1e3); tot.append((t2 - t0) / burst * 1e3)
    enq.sort(); tot.sort()
    print("loader in the loop (PGNN_LOADER_PREFETCH=%s): host enqueue %.3f ms/step (median), burst wall %.3f ms/step (median)" % (
        os.environ.get("PGNN_LOADER_PREFETCH", "1"), enq[6], tot[6]))
