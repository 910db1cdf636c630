"""A/B of the one-call network with BatchNorm statistics from the GEMM epilogue (PGNN_BN_STATS_IN_GEMM=1) vs the separate pass (=0):
per-parameter gradient differences between the two, and of each against the float64 CPU oracle on the same weights and batch.
usage: python tools/bn_stats_ab.py [graphs=256]"""
import copy, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import chem as ochem
from pretrain_gnns_amd import ops
from pretrain_gnns_amd.chem import model as hchem
from pretrain_gnns_amd.data import synthetic
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(8)
ref = ochem.GNN(5, 300)
d = synthetic.chem_masking_batch(graphs, seed=9).to("cpu")  # (device-collated; the float64 reference runs on the host)
w = torch.randn(d.x.size(0), 300)
ref64 = copy.deepcopy(ref).double()
ref64.train()
out64 = ref64(d.x, d.edge_index, d.edge_attr)
(out64 * w.double()).sum().backward()
g64 = {k: p.grad.clone() for k, p in ref64.named_parameters()}
res = {}
for flag in ("1", "0"):
    os.environ["PGNN_BN_STATS_IN_GEMM"] = flag
    ops.load().pgnn_reload_env()
    m = hchem.GNN(5, 300)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().train()
    dd = d.clone().to("cuda")
    out = m(dd.x, dd.edge_index, dd.edge_attr)
    (out * w.cuda()).sum().backward()
    res[flag] = (out.detach().double().cpu(), {k: p.grad.double().cpu() for k, p in m.named_parameters()})
print("out: |1-0| max %.3e   |1-f64| max %.3e   |0-f64| max %.3e" % ((res["1"][0] - res["0"][0]).abs().max(), (res["1"][0] - out64.detach()).abs().max(),
                                                                    (res["0"][0] - out64.detach()).abs().max()))
print("%-34s %10s %10s %10s %10s %10s" % ("param", "scale", "relL2 1-0", "relL2 1-64", "relL2 0-64", "max 1-0"))
for k in g64:
    a, b, t = res["1"][1][k], res["0"][1][k], g64[k]
    nrm = float(t.norm()) + 1e-30
    print("%-34s %10.3e %10.2e %10.2e %10.2e %10.2e" % (k, float(t.abs().max()), float((a - b).norm()) / nrm, float((a - t).norm()) / nrm,
                                                       float((b - t).norm()) / nrm, float((a - b).abs().max())))
