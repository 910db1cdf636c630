"""fixed vs per-k-step cost of the forward GEMM at small M: time(K) = a + b*K"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pretrain_gnns_amd import ops
lib, sp, dev = ops.load(), ops.stream_ptr(), "cuda"
def timeit(fn, iters=50):
    for _ in range(5): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e3
for m in (6747, 16384):
    for n in (304, 608):
        row = []
        for k in (16, 64, 128, 256, 512, 1024, 2048):
            x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev); y = torch.empty(m, n, device=dev)
            us = timeit(lambda: ops.check(lib.pgnn_linear_fwd(x.data_ptr(), k, w.data_ptr(), b.data_ptr(), y.data_ptr(), n, m, k, n, 1, sp), "f"))
            row.append("K=%d: %.1f us (%.0f TF)" % (k, us, 2.0 * m * k * n / us / 1e6))
        print("M=%d N=%d cfg=%s | " % (m, n, os.environ.get("PGNN_GEMM_CFG", "auto")) + " | ".join(row))
