"""engine-side cost of a custom Function that returns gradients for K leaf parameters (no kernels at all)."""
import sys, time, torch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 47
dev = "cuda"
params = [torch.nn.Parameter(torch.zeros(300, device=dev)) for _ in range(K)]
flat = torch.zeros(K * 300, device=dev)
pieces = None

class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *ps):
        return x.clone()
    @staticmethod
    def backward(ctx, g):
        return (None,) + flat.split_with_sizes([300] * K)

class G(torch.autograd.Function):  # writes .grad itself, returns nothing for the parameters
    @staticmethod
    def forward(ctx, x, *ps):
        ctx.ps = ps
        return x.clone()
    @staticmethod
    def backward(ctx, g):
        for p, t in zip(ctx.ps, flat.split_with_sizes([300] * K)):
            p.grad = t
        return (None,) * (K + 1)

x = torch.zeros(8, device=dev, requires_grad=True)
for name, fn in (("47 returned grads" if K == 47 else "%d returned grads" % K, F), ("grads written directly", G)):
    for _ in range(20):
        for p in params: p.grad = None
        fn.apply(x, *params).sum().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 300
    for _ in range(n):
        for p in params: p.grad = None
        out = fn.apply(x, *params).sum()
        t1 = time.perf_counter()
        out.backward()
    torch.cuda.synchronize()
    print("%-26s %.1f us per fwd+bwd" % (name, (time.perf_counter() - t0) / n * 1e6))

class H(torch.autograd.Function):  # parameters not passed through apply at all
    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return x.clone()
    @staticmethod
    def backward(ctx, g):
        for p, t in zip(ctx.holder, flat.split_with_sizes([300] * K)):
            p.grad = t
        return None, None

for _ in range(20):
    H.apply(x, params).sum().backward()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 300
for _ in range(n):
    for p in params: p.grad = None
    H.apply(x, params).sum().backward()
torch.cuda.synchronize()
print("%-26s %.1f us per fwd+bwd" % ("params outside apply", (time.perf_counter() - t0) / n * 1e6))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    for p in params: p.grad = None
torch.cuda.synchronize()
print("%-26s %.1f us" % ("(clearing .grad alone)", (time.perf_counter() - t0) / n * 1e6))
