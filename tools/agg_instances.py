"""HIP-event timing of the chem aggregation instances on the roofline batch: plain (layer 0) and BatchNorm-on-read (layers 1-4).
usage: [ORDER=smiles|survey] [RELABEL=0|1] python tools/agg_instances.py [graphs=16384] [launches=50]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops
from pretrain_gnns_amd.data import synthetic

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = "cuda"
big, info = synthetic.chem_aggregation_batch(graphs, os.environ.get("ORDER", "smiles"), os.environ.get("RELABEL", "1") != "0", device=dev)
n, e = big.x.size(0), big.edge_index.size(1)
g = ops.build_chem_graph(big.edge_index, big.edge_attr, n)
torch.manual_seed(0)
z = torch.randn(n, 300, device=dev)
coef = torch.stack([torch.rand(300, device=dev) + 0.5, torch.randn(300, device=dev) * 0.2]).contiguous()
out = torch.empty_like(z)
e1, e2 = torch.randn(6, 300, device=dev), torch.randn(3, 300, device=dev)
lib, sp = ops.load(), ops.stream_ptr()
alg = 2400.0 * n + 6.0 * e + 4.0 * (n + 1)

def plain():
    ops.check(lib.pgnn_chem_aggregate_fwd(z.data_ptr(), 300, g.in_ptr.data_ptr(), g.in_src.data_ptr(), g.in_code.data_ptr(),
                                          e1.data_ptr(), e2.data_ptr(), None, out.data_ptr(), 300, n, 300, sp), "agg")
def bn_on_read():
    ops.check(lib.pgnn_chem_aggregate_bn_fwd(z.data_ptr(), 300, coef.data_ptr(), 1, g.in_ptr.data_ptr(), g.in_src.data_ptr(),
                                             g.in_code.data_ptr(), e1.data_ptr(), e2.data_ptr(), out.data_ptr(), 300, n, 300, sp), "agg_bn")
def time(fn):
    import time as _t
    t0 = _t.perf_counter()
    while _t.perf_counter() - t0 < 0.1: fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), ms[0], ms[-1]

print("batch:", info, "nodes", n, "edges", e, "algorithmic bytes", int(alg))
for rep in range(2):
    for name, fn in (("plain", plain), ("bn_on_read", bn_on_read)):
        mean, lo, hi = time(fn)
        print("%-11s %.1f us (min %.1f max %.1f)  %.0f GB/s = %.3f of 8 TB/s" % (name, mean * 1e3, lo * 1e3, hi * 1e3, alg / mean / 1e6, alg / mean / 8e9))
if os.environ.get("CHECK", "1") != "0":
    y = torch.clamp_min(torch.addcmul(coef[1], coef[0], z), 0.0)  # fmaf? torch.addcmul is not guaranteed fused: compare against the materialising kernel instead
    ws = torch.empty(int(lib.pgnn_bn_workspace_bytes(n, 300)), dtype=torch.uint8, device=dev)
    ops.check(lib.pgnn_bn_apply_fwd(z.data_ptr(), 300, coef.data_ptr(), 1, y.data_ptr(), 300, 0.0, 0, n, 300, sp), "apply")
    bn_on_read(); got = out.clone()
    zz = z; z = y; plain(); z = zz
    print("bn_on_read bit-identical to apply-then-aggregate:", torch.equal(got, out))
