"""cProfile of the Python side of the model forward + backward (host overhead outside the C calls)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd.chem import model as hmodel
from pretrain_gnns_amd.data import synthetic
dev = "cuda"
batch = synthetic.chem_masking_batch(256, seed=7).to(dev)
torch.manual_seed(0)
m = hmodel.GNN(5, 300).to(dev)
w = torch.randn(batch.x.size(0), 300, device=dev)
def step():
    m.zero_grad()
    out = m(batch.x, batch.edge_index, batch.edge_attr)
    (out * w).sum().backward()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
