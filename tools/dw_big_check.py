"""the large weight gradient (pgnn_linear_bwd_weight at >= 24 576 rows) under PGNN_DW_BIG_TILE = 320 | 160 | 128: HIP-event time of both
products of a chem layer and their error against float64 on a sample of entries.  usage: python tools/dw_big_check.py [rows=438792]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pretrain_gnns_amd import ops
m = int(sys.argv[1]) if len(sys.argv) > 1 else 438792
dev = "cuda"
lib, sp = ops.load(), ops.stream_ptr()
torch.manual_seed(0)
d = 300
dz, hid = torch.randn(m, d, device=dev) * 1e-3, torch.relu(torch.randn(m, 2 * d, device=dev))
dhid, agg = torch.randn(m, 2 * d, device=dev) * 1e-3, torch.randn(m, d, device=dev)
nb = lambda k, n: int(lib.pgnn_linear_bwd_weight_workspace_bytes(m, k, n))
ws = torch.empty(max(nb(2 * d, d), nb(d, 2 * d)), dtype=torch.uint8, device=dev)
rows = torch.randint(0, d, (64,), device=dev)
want2 = (dz[:, rows].double().t() @ hid.double())  # [64, 600]
scale2 = (dz[:, rows].double().abs().t() @ hid.double().abs())
res = {}
for tile in ("320", "160", "128"):
    os.environ["PGNN_DW_BIG_TILE"] = tile
    lib.pgnn_reload_env()
    dw2, db2 = torch.empty(d, 2 * d, device=dev), torch.empty(d, device=dev)
    dw1, db1 = torch.empty(2 * d, d, device=dev), torch.empty(2 * d, device=dev)
    def run():
        ops.check(lib.pgnn_linear_bwd_weight(dz.data_ptr(), d, hid.data_ptr(), 2 * d, dw2.data_ptr(), db2.data_ptr(), m, 2 * d, d, ws.data_ptr(), ws.numel(), sp), "dw2")
        ops.check(lib.pgnn_linear_bwd_weight(dhid.data_ptr(), 2 * d, agg.data_ptr(), d, dw1.data_ptr(), db1.data_ptr(), m, d, 2 * d, ws.data_ptr(), ws.numel(), sp), "dw1")
    for _ in range(3): run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    err = ((dw2[rows].double() - want2).abs() / scale2).max().item()
    eb = ((db2.double() - dz.double().sum(0)).abs() / dz.double().abs().sum(0)).max().item()
    res[tile] = (dw2.clone(), dw1.clone())
    print("PGNN_DW_BIG_TILE=%s: %.1f us per PAIR of products (incl. the folds), dW2 max err / bound %.2e, db2 %.2e" % (tile, s.elapsed_time(e) / 10 * 1e3, err, eb))
for t in ("160", "128"):
    print("max |dW(%s) - dW(320)| / max|dW|: %.2e %.2e" % (t, ((res[t][0] - res["320"][0]).abs().max() / res["320"][0].abs().max()).item(),
                                                       ((res[t][1] - res["320"][1]).abs().max() / res["320"][1].abs().max()).item()))
