"""does a 128-byte-aligned leading dimension of the activations help the k-major DMA pieces?
usage: python tools/gemm_ld_bench.py [rows]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops
m = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
lib, sp, dev = ops.load(), ops.stream_ptr(), "cuda"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters
for (k, n) in ((300, 600), (600, 300)):
    for pad in (0, 1):
        ldk = (k + 31) // 32 * 32 if pad else k
        ldn = (n + 31) // 32 * 32 if pad else n
        torch.manual_seed(0)
        xb = torch.randn(m, ldk, device=dev); w = torch.randn(n, k, device=dev) * 0.05; b = torch.randn(n, device=dev)
        yb = torch.empty(m, ldn, device=dev); dyb = torch.randn(m, ldn, device=dev); dxb = torch.empty(m, ldk, device=dev)
        dw = torch.empty(n, k, device=dev); db = torch.empty(n, device=dev)
        ws = torch.empty(int(lib.pgnn_linear_bwd_weight_workspace_bytes(m, k, n)), dtype=torch.uint8, device=dev)
        fl = 2.0 * m * k * n
        f = timeit(lambda: ops.check(lib.pgnn_linear_fwd(xb.data_ptr(), ldk, w.data_ptr(), b.data_ptr(), yb.data_ptr(), ldn, m, k, n, 1, sp), "f"))
        d = timeit(lambda: ops.check(lib.pgnn_linear_bwd_data(dyb.data_ptr(), ldn, w.data_ptr(), None, 0, dxb.data_ptr(), ldk, m, k, n, sp), "d"))
        g = timeit(lambda: ops.check(lib.pgnn_linear_bwd_weight(dyb.data_ptr(), ldn, xb.data_ptr(), ldk, dw.data_ptr(), db.data_ptr(), m, k, n, ws.data_ptr(), ws.numel(), sp), "w"))
        print("M=%d K=%d N=%d ld(x)=%d ld(y)=%d  fwd %.1f us %.1f TF | bwd_data %.1f us %.1f TF | bwd_weight %.1f us %.1f TF"
              % (m, k, n, ldk, ldn, f * 1e3, fl / f / 1e9, d * 1e3, fl / d / 1e9, g * 1e3, fl / g / 1e9))
