"""the paired weight gradients of a chem layer (pgnn_linear_bwd_weight_pair) ALONE at the 256-graph batch's row count: three bf16 planes
(default) against two fp16 planes under column scales (PGNN_DW_2P=1), HIP events around the whole call (column maxima, product, fold).
usage: python tools/dw_pair_check.py [rows=6740]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pretrain_gnns_amd import ops
m = int(sys.argv[1]) if len(sys.argv) > 1 else 6740
dev = "cuda"
lib, sp = ops.load(), ops.stream_ptr()
torch.manual_seed(0)
d = 300
dz, hid = torch.randn(m, d, device=dev) * 1e-3, torch.relu(torch.randn(m, 2 * d, device=dev))
dhid, agg = torch.randn(m, 2 * d, device=dev) * 1e-3, torch.randn(m, d, device=dev)
nb = lambda k, n: int(lib.pgnn_linear_bwd_weight_workspace_bytes(m, k, n))
ws = torch.empty(nb(2 * d, d) + nb(d, 2 * d), dtype=torch.uint8, device=dev)
dw2, db2 = torch.empty(d, 2 * d, device=dev), torch.empty(d, device=dev)
dw1, db1 = torch.empty(2 * d, d, device=dev), torch.empty(2 * d, device=dev)
def run():
    ops.check(lib.pgnn_linear_bwd_weight_pair(dz.data_ptr(), d, hid.data_ptr(), 2 * d, dw2.data_ptr(), db2.data_ptr(), 2 * d, d,
                                              dhid.data_ptr(), 2 * d, agg.data_ptr(), d, dw1.data_ptr(), db1.data_ptr(), d, 2 * d, m,
                                              ws.data_ptr(), ws.numel(), sp), "pair")
for rep in range(3):
    for flag in ("0", "1"):
        os.environ["PGNN_DW_2P"] = flag
        lib.pgnn_reload_env()
        for _ in range(20): run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(200): run()
        e.record(); torch.cuda.synchronize()
        print("PGNN_DW_2P=%s: %.1f us per call (%s)" % (flag, s.elapsed_time(e) / 200 * 1e3, "two fp16 planes + column maxima" if flag == "1" else "three bf16 planes"))
