"""Soak of the prefetching resident loader (round 6): `epochs` epochs of masking training over a 4 096-molecule dataset with the loader
collating a batch ahead on its side stream, and again collating in line; the two runs must end in bit-identical parameters and loss
sums (a race between the side stream's writes and the consumer's reads, or a recycled block written too early, would show).
usage: python tools/soak_loader.py [epochs=25] [batch=256]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pretrain_gnns_amd import train as steps
from pretrain_gnns_amd.data import resident, synthetic

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 25
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)
graphs = [synthetic.zinc_like_graph(rng) for _ in range(4096)]
ds = resident.ResidentDataset.from_graphs(graphs, dev)
out = {}
for flag in os.environ.get("ORDER", "1,0,1,0").split(","):
    os.environ["PGNN_LOADER_PREFETCH"] = flag
    loader = resident.ResidentLoader(ds, bs, shuffle=True, seed=3, mask_rate=0.15, drop_last=False)
    mods = bench.make_models(dev)
    opts = bench.make_optimizers(mods)
    acc = steps.epoch_accumulator(dev)
    t0 = time.perf_counter()
    n = 0
    for _ in range(epochs):
        for b in loader:
            steps.chem_masking_step(mods, opts, b, readback="epoch", accum=acc)
            n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if flag in out:
        assert out[flag][0] == acc.cpu().tolist()
    out[flag] = (acc.cpu().tolist(), [p.detach().clone() for m in mods for p in m.parameters()])
    print("PGNN_LOADER_PREFETCH=%s: %d steps, %.3f ms/step, sums %s" % (flag, n, dt / n * 1e3, out[flag][0]))
same = out["1"][0] == out["0"][0] and all(torch.equal(a, b) for a, b in zip(out["1"][1], out["0"][1]))
print("bit-identical:", same)
sys.exit(0 if same else 1)
