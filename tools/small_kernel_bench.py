"""the small reductions of a 256-graph train step (N ~ 6.7k rows, D = 300) one by one: BatchNorm statistics / backward,
bond-table gradient (rowfeat_matmul_bwd), split-K weight gradient, atom-embedding gradient.  HIP-event time per call in
steady state, for A/B of their launch geometry (PGNN_BN_ROWS_PER_BLOCK, PGNN_ROWFEAT_ROWS_PER_BLOCK).
usage: python tools/small_kernel_bench.py [rows=6747]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6747
dev, d = "cuda", 300
lib, sp = ops.load(), ops.stream_ptr()
def timeit(fn, iters=50, warm=0.05):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
torch.manual_seed(0)
x = torch.randn(n, d, device=dev); dy = torch.randn(n, d, device=dev); y = torch.empty_like(x); dx = torch.empty_like(x)
gamma, beta = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
rm, rv = torch.zeros(d, device=dev), torch.ones(d, device=dev)
sm, si = torch.empty(d, device=dev), torch.empty(d, device=dev)
dg, db = torch.empty(d, device=dev), torch.empty(d, device=dev)
coef = torch.empty(2, d, device=dev)
cfeat = torch.rand(n, 9, device=dev)
gt = torch.empty(9, d, device=dev)
def run(tag):
    ws = torch.empty(int(lib.pgnn_bn_workspace_bytes(n, d)), dtype=torch.uint8, device=dev)
    t_stats = timeit(lambda: ops.check(lib.pgnn_bn_stats_fwd(x.data_ptr(), d, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, 1,
                                                              sm.data_ptr(), si.data_ptr(), coef.data_ptr(), n, d, ws.data_ptr(), ws.numel(), sp), "stats"))
    t_fwd = timeit(lambda: ops.check(lib.pgnn_bn_fwd(x.data_ptr(), d, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, 1, 1, y.data_ptr(), d,
                                                     sm.data_ptr(), si.data_ptr(), 0.0, 0, n, d, ws.data_ptr(), ws.numel(), sp), "fwd"))
    t_bwd = timeit(lambda: ops.check(lib.pgnn_bn_bwd(dy.data_ptr(), d, x.data_ptr(), d, gamma.data_ptr(), beta.data_ptr(), sm.data_ptr(), si.data_ptr(), 1, 1, dx.data_ptr(), d,
                                                     dg.data_ptr(), db.data_ptr(), 0.0, 0, n, d, ws.data_ptr(), ws.numel(), sp), "bwd"))
    ws2 = torch.empty(int(lib.pgnn_rowfeat_matmul_bwd_workspace_bytes(n, 9, d)), dtype=torch.uint8, device=dev)
    t_rf = timeit(lambda: ops.check(lib.pgnn_rowfeat_matmul_bwd(cfeat.data_ptr(), 9, dy.data_ptr(), d, gt.data_ptr(), d, n, d, ws2.data_ptr(), ws2.numel(), sp), "rf"))
    print("%-28s N=%d | bn stats %.1f us | bn fwd (stats+apply) %.1f us | bn bwd %.1f us | rowfeat bwd (kc 9) %.1f us" % (tag, n, t_stats, t_fwd, t_bwd, t_rf), flush=True)
    return dx.clone(), dg.clone(), gt.clone()
def setenv(**kv):
    for k, v in kv.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    lib.pgnn_reload_env()
ref = None
for bn_rows, rf_rows in ((32, 64), (16, 32), (8, 16), (4, 8), (8, 8)):
    setenv(PGNN_BN_ROWS_PER_BLOCK=bn_rows, PGNN_ROWFEAT_ROWS_PER_BLOCK=rf_rows)
    out = run("bn rows/block %d, rowfeat %d" % (bn_rows, rf_rows))
    if ref is None: ref = out
    else: print("   max rel diff vs first config:", [float(((a - b).abs().max() / b.abs().max())) for a, b in zip(out, ref)])
setenv(PGNN_BN_ROWS_PER_BLOCK=None, PGNN_ROWFEAT_ROWS_PER_BLOCK=None)
# weight gradient (split-K) and the two data products, for reference
m = n
for (k, nn) in ((300, 600), (600, 300)):
    xx = torch.randn(m, k, device=dev); dyy = torch.randn(m, nn, device=dev); dw = torch.empty(nn, k, device=dev); dbb = torch.empty(nn, device=dev)
    ws3 = torch.empty(int(lib.pgnn_linear_bwd_weight_workspace_bytes(m, k, nn)), dtype=torch.uint8, device=dev)
    t = timeit(lambda: ops.check(lib.pgnn_linear_bwd_weight(dyy.data_ptr(), nn, xx.data_ptr(), k, dw.data_ptr(), dbb.data_ptr(), m, k, nn, ws3.data_ptr(), ws3.numel(), sp), "w"))
    print("weight gradient + bias (split-K + reduce) M=%d K=%d N=%d: %.1f us" % (m, k, nn, t))
