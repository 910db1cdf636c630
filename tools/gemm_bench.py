"""micro-benchmark of the fp32 MFMA GEMM entry points (HIP-event timing).
usage: [PGNN_GEMM_CFG=c] python tools/gemm_bench.py [rows ...]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops
rows = [int(a) for a in sys.argv[1:]] or [6747, 262144]
lib, sp, dev = ops.load(), ops.stream_ptr(), "cuda"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters
for m in rows:
    torch.manual_seed(0)
    for (k, n) in ((300, 600), (600, 300)):
        x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05; b = torch.randn(n, device=dev)
        y = torch.empty(m, n, device=dev); dy = torch.randn(m, n, device=dev); dx = torch.empty(m, k, device=dev)
        dw = torch.empty(n, k, device=dev); db = torch.empty(n, device=dev)
        ws = torch.empty(int(lib.pgnn_linear_bwd_weight_workspace_bytes(m, k, n)), dtype=torch.uint8, device=dev)
        fl = 2.0 * m * k * n
        f = timeit(lambda: ops.check(lib.pgnn_linear_fwd(x.data_ptr(), k, w.data_ptr(), b.data_ptr(), y.data_ptr(), n, m, k, n, 1, sp), "f"))
        d = timeit(lambda: ops.check(lib.pgnn_linear_bwd_data(dy.data_ptr(), n, w.data_ptr(), None, 0, dx.data_ptr(), k, m, k, n, sp), "d"))
        g = timeit(lambda: ops.check(lib.pgnn_linear_bwd_weight(dy.data_ptr(), n, x.data_ptr(), k, dw.data_ptr(), db.data_ptr(), m, k, n, ws.data_ptr(), ws.numel(), sp), "w"))
        t = timeit(lambda: torch.addmm(b, x, w.t(), out=y))
        ref = torch.relu(torch.addmm(b, x, w.t()))
        ops.check(lib.pgnn_linear_fwd(x.data_ptr(), k, w.data_ptr(), b.data_ptr(), y.data_ptr(), n, m, k, n, 1, sp), "f")
        err = (y - ref).abs().max().item()
        print("cfg %s M=%d K=%d N=%d  fwd %.1f us %.1f TF | bwd_data %.1f us %.1f TF | bwd_weight(+db) %.1f us %.1f TF | rocBLAS addmm %.1f us %.1f TF | err %.1e"
              % (os.environ.get("PGNN_GEMM_CFG", "auto"), m, k, n, f * 1e3, fl / f / 1e9, d * 1e3, fl / d / 1e9, g * 1e3, fl / g / 1e9, t * 1e3, fl / t / 1e9, err))
