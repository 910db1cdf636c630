"""time pgnn_linear_fwd_2p at large row counts: the resident-plane kernel (k_gemm2pr, PGNN_GEMM2P_RES) against the tiled one
usage: python tools/gemm2p_large.py [rows ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import steady_state_ms  # noqa: E402
from pretrain_gnns_amd import ops  # noqa: E402


def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [262144, 65536, 32768, 16384]
    dev = torch.device("cuda:0")
    lib = ops.load()
    torch.manual_seed(0)
    w1, b1 = torch.randn(600, 300, device=dev) * 0.05, torch.randn(600, device=dev)
    w2, b2 = torch.randn(300, 600, device=dev) * 0.05, torch.randn(300, device=dev)
    p1, p2 = ops.weight_planes_2p([w1, w2])
    for rows in rows_list:
        x = torch.randn(rows, 300, device=dev)
        hid = torch.empty(rows, 600, device=dev)
        z = torch.empty(rows, 300, device=dev)
        amax = torch.zeros(rows, dtype=torch.int32, device=dev)
        xam = x.abs().max(dim=1).values.contiguous().view(torch.int32)
        ops.linear_fwd_2p(x, p1, b1, 600, relu=True, out=hid, y_amax=amax)
        flops = 2.0 * rows * 300 * 600
        for knob, extra in (("0", {}), ("2", {}), ("0", {}), ("2", {})):
            os.environ["PGNN_GEMM2P_RES"] = knob
            for kk in ("PGNN_GEMM2PR_FLAGS", "PGNN_GEMM2PR_NW", "PGNN_GEMM2PR_PF"):
                os.environ.pop(kk, None)
            os.environ.update(extra)
            lib.pgnn_reload_env()
            line = "rows %7d RES=%s %s" % (rows, knob, " ".join("%s=%s" % kv for kv in extra.items()))
            for tag, fn in (("300->600 self", lambda: ops.linear_fwd_2p(x, p1, b1, 600, relu=True, out=hid)),
                            ("300->600 given", lambda: ops.linear_fwd_2p(x, p1, b1, 600, relu=True, out=hid, x_amax=xam)),
                            ("600->300 given", lambda: ops.linear_fwd_2p(hid, p2, b2, 300, out=z, x_amax=amax))):
                ms, per, iters = steady_state_ms(fn, warm_s=0.05, iters=30)
                line += " | %s %.1f us %.0f TF" % (tag, ms * 1e3, flops / (ms * 1e-3) / 1e12)
            print(line, flush=True)
    del os.environ["PGNN_GEMM2P_RES"]
    for kk in ("PGNN_GEMM2PR_FLAGS", "PGNN_GEMM2PR_NW", "PGNN_GEMM2PR_PF"):
        os.environ.pop(kk, None)
    lib.pgnn_reload_env()


if __name__ == "__main__":
    main()
