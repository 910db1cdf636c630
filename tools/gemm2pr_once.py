"""a handful of launches of the 300 -> 600 forward product at a large row count (for rocprofv3 --pmc passes)
usage: python tools/gemm2pr_once.py [rows] [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pretrain_gnns_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
w1, b1 = torch.randn(600, 300, device=dev) * 0.05, torch.randn(600, device=dev)
(p1,) = ops.weight_planes_2p([w1])
x = torch.randn(rows, 300, device=dev)
hid = torch.empty(rows, 600, device=dev)
xam = x.abs().max(dim=1).values.contiguous().view(torch.int32)
for _ in range(n):
    ops.linear_fwd_2p(x, p1, b1, 600, relu=True, out=hid, x_amax=xam)
torch.cuda.synchronize()
print("done", rows, n)
