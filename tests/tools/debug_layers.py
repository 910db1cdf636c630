"""debug aid: per-layer forward/backward comparison of the HIP stack against the fp64 oracle."""
import copy, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import chem as ochem
from pretrain_gnns_amd.chem import model as hchem
from pretrain_gnns_amd.data import synthetic
from pretrain_gnns_amd import ops

torch.manual_seed(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ref = ochem.GNN(5, 300)
hip = hchem.GNN(5, 300); hip.load_state_dict(ref.state_dict()); hip = hip.cuda()
ref64 = copy.deepcopy(ref).double()
b = synthetic.chem_masking_batch(G, seed=G); d = b.clone().to("cuda")

def run(model, x, ei, ea, dev, dt):
    acts, grads = {}, {}
    hs = []
    def mk(name):
        def hook(mod, inp, out):
            acts[name] = out.detach().cpu().double()
            out.register_hook(lambda g, n=name: grads.__setitem__(n, g.detach().cpu().double()))
        return hook
    for i, (c, bn) in enumerate(zip(model.gnns, model.batch_norms)):
        hs.append(c.register_forward_hook(mk("conv%d" % i)))
        if isinstance(bn, torch.nn.BatchNorm1d) and dev == "cpu":
            hs.append(bn.register_forward_hook(mk("bn%d" % i)))
    out = model(x, ei, ea)
    torch.manual_seed(1)
    w = torch.randn(out.shape, dtype=torch.float64).to(dt).to(dev)
    (out * w).sum().backward()
    for h in hs: h.remove()
    return out.detach().cpu().double(), acts, grads

o64, a64, g64 = run(ref64, b.x, b.edge_index, b.edge_attr, "cpu", torch.float64)
o32, a32, g32 = run(ref, b.x, b.edge_index, b.edge_attr, "cpu", torch.float32)
oh, ah, gh = run(hip, d.x, d.edge_index, d.edge_attr, "cuda", torch.float32)
rel = lambda a, t: ((a - t).abs().max() / (t.abs().max() + 1e-30)).item()
print("out  cpu32 %.2e  hip %.2e" % (rel(o32, o64), rel(oh, o64)))
for k in sorted(a64):
    if k in ah:
        print("act  %-6s cpu32 %.2e hip %.2e | grad cpu32 %.2e hip %.2e" % (k, rel(a32[k], a64[k]), rel(ah[k], a64[k]), rel(g32[k], g64[k]), rel(gh[k], g64[k])))
for (n, p64), (_, p32), (_, ph) in zip(ref64.named_parameters(), ref.named_parameters(), hip.named_parameters()):
    e32, eh = rel(p32.grad.double(), p64.grad), rel(ph.grad.cpu().double(), p64.grad)
    flag = "  <<<" if eh > 20 * e32 + 1e-5 else ""
    print("%-34s cpu32 %.2e hip %.2e%s" % (n, e32, eh, flag))
