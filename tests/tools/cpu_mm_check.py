"""debug aid: is torch-CPU fp32 matmul/backward accurate on this host? (vs fp64)"""
import torch, os
print(torch.__version__, "threads", torch.get_num_threads(), torch.get_float32_matmul_precision())
try:
    print("mkldnn matmul fp32_precision", torch.backends.mkldnn.matmul.fp32_precision)
except Exception as e:
    print("n/a", e)
torch.manual_seed(0)
rel = lambda a, t: ((a.double() - t).abs().max() / t.abs().max()).item()
for th in (None, 16, 1):
    if th: torch.set_num_threads(th)
    for (m, k, n) in [(850, 300, 600), (850, 600, 300), (26, 300, 600)]:
        a, b = torch.randn(m, k), torch.randn(k, n)
        print("threads", torch.get_num_threads(), (m, k, n), "mm %.2e" % rel(a @ b, a.double() @ b.double()),
              "mm(a, w.t()) %.2e" % rel(a @ b.t().contiguous().t(), a.double() @ b.double()),
              "addmm-linear %.2e" % rel(torch.nn.functional.linear(a, b.t().contiguous()), a.double() @ b.double()),
              "mm(a.t, ) %.2e" % rel(a.t() @ torch.randn(m, 7), a.t().double() @ torch.randn(m, 7).double() * 0 + a.t().double() @ torch.zeros(m,7).double()) if False else "")
        g = torch.randn(m, n)
        x = a.clone().requires_grad_(True); w = b.t().contiguous().clone().requires_grad_(True)
        torch.nn.functional.linear(x, w).backward(g)
        print("    dX %.2e dW %.2e" % (rel(x.grad, g.double() @ w.detach().double()), rel(w.grad, g.double().t() @ a.double())))
