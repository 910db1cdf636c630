"""CPU study for DESIGN 8.1: how accurate is a product on TWO fp16 planes per operand (three MFMA products per accumulator,
power-of-two scale per row) against the THREE bf16 planes (six products) the kernels run now, a plain fp32 product, and float64?
Emulation: planes by round-to-nearest casts, every plane product exact in fp32 (8 x 8 and 11 x 11 significant bits), accumulation
in fp32 by torch's CPU matmul (a different order than the MFMA's, the same kind of error).  Shapes and value ranges of the GIN mlp:
activations with row magnitudes spread over six decades (forward), gradients ~1e-6 (backward).
usage: python tools/two_plane_numerics.py"""
import torch

torch.manual_seed(0)


def split_bf16x3(x):
    h = x.to(torch.bfloat16).float()
    m = (x - h).to(torch.bfloat16).float()
    l = (x - h - m).to(torch.bfloat16).float()
    return h, m, l


def row_scale(x, top=14):
    """2^k per row such that the row's largest magnitude lands in [2^(top-1), 2^top)"""
    amax = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    return torch.exp2(top - 1 - torch.floor(torch.log2(amax)))


def split_fp16x2(x, scaled=True):
    s = row_scale(x) if scaled else torch.ones(x.size(0), 1)
    y = x * s
    h = y.to(torch.float16).float()
    l = (y - h).to(torch.float16).float()
    return h, l, s


def prod_bf16x3(a, w):
    a1, a2, a3 = split_bf16x3(a)
    w1, w2, w3 = split_bf16x3(w)
    return (a1 @ w3.t() + a2 @ w2.t() + a3 @ w1.t()) + (a1 @ w2.t() + a2 @ w1.t()) + a1 @ w1.t()


def prod_fp16x2(a, w, scaled=True):
    a1, a2, sa = split_fp16x2(a, scaled)
    w1, w2, sw = split_fp16x2(w, scaled)
    return ((a1 @ w2.t() + a2 @ w1.t()) + a1 @ w1.t()) / (sa * sw.t())


def report(name, a, w):
    truth = a.double() @ w.double().t()
    denom = (a.double().abs() @ w.double().abs().t()).clamp_min(1e-300)  # componentwise backward-error scale
    rows = []
    for label, c in (("fp32 matmul", a @ w.t()), ("three bf16 planes, six products", prod_bf16x3(a, w)),
                     ("two fp16 planes, three products, row scales", prod_fp16x2(a, w)),
                     ("two fp16 planes, three products, NO scale", prod_fp16x2(a, w, scaled=False))):
        err = ((c.double() - truth).abs() / denom)
        rel_out = (c.double() - truth).abs().max() / truth.abs().max()
        rows.append((label, float(err.max()), float(err.mean()), float(rel_out)))
    print(name)
    for label, emax, emean, rel in rows:
        print("  %-46s max |err| / (|a| |w|) %.2e   mean %.2e   max |err| / max |c| %.2e" % (label, emax, emean, rel))


m, k, n = 4096, 300, 600
w = (torch.rand(n, k) * 2 - 1) / k ** 0.5
act = torch.randn(m, k) * torch.exp(torch.randn(m, 1) * 2.0)  # row magnitudes over ~6 decades
report("forward 300 -> 600, activations with row magnitudes over six decades", act, w)
report("forward 600 -> 300, post-ReLU activations", torch.relu(torch.randn(m, n)) * 3.0, (torch.rand(k, n) * 2 - 1) / n ** 0.5)
grad = torch.randn(m, n) * 1e-6 * torch.exp(torch.randn(m, 1))
report("backward-data 600 -> 300, gradients ~1e-6", grad, w.t().contiguous())
# weight gradient: the contraction runs over the ROWS, so the scales belong to the columns of both operands
report("weight gradient (6 740 rows contracted), gradients ~1e-6 x activations", (torch.randn(6740, n) * 1e-6).t().contiguous(),
       (torch.randn(6740, k) * torch.exp(torch.randn(6740, 1))).t().contiguous())
