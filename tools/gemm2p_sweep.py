"""tiled (k_gemm2pw) against resident-plane (k_gemm2pr) two-plane forward products over row counts and the GIN mlp's shapes
usage: python tools/gemm2p_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import steady_state_ms  # noqa: E402
from pretrain_gnns_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = ops.load()
    torch.manual_seed(0)
    for k, n in ((300, 600), (600, 300), (600, 600)):
        w, b = torch.randn(n, k, device=dev) * 0.05, torch.randn(n, device=dev)
        (pl,) = ops.weight_planes_2p([w])
        for rows in (4096, 5500, 6740, 8192, 10249, 12288, 16384, 24576, 32768):
            x = torch.randn(rows, k, device=dev)
            y = torch.empty(rows, n, device=dev)
            xam = x.abs().max(dim=1).values.contiguous().view(torch.int32)
            line = "K %3d N %3d rows %6d" % (k, n, rows)
            for knob in ("0", "2", "0", "2"):
                os.environ["PGNN_GEMM2P_RES"] = knob
                lib.pgnn_reload_env()
                ms, per, iters = steady_state_ms(lambda: ops.linear_fwd_2p(x, pl, b, n, relu=True, out=y, x_amax=xam), warm_s=0.03, iters=40)
                line += " | RES=%s %.1f us" % (knob, ms * 1e3)
            print(line, flush=True)
    del os.environ["PGNN_GEMM2P_RES"]
    lib.pgnn_reload_env()


if __name__ == "__main__":
    main()
