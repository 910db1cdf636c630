"""Data parallelism for the pre-training hot path: shard by graph, one collective per step.

The reference is single-device (chem/pretrain_masking.py:114); this layer is new design, not a
translation.  A batch is a block-diagonal union of independent graphs (chem/batch.py:31-52), so
the path shards with no data-path exchange: every rank (one process per GPU) collates its own
graphs, runs forward/backward locally, and the only collective is ONE sum-all-reduce per step of
a single flat fp32 bucket holding every gradient (GNN 7.43 MB + heads; RCCL over xGMI picks
direct reduce-scatter/all-gather on the fully connected 8-GPU node), followed by identical Adam
updates on every rank.  BatchNorm statistics stay per rank (standard DDP semantics).

The layer is model-agnostic (any nn.Module, any torch.distributed backend), which is how the
world_size-2 ``gloo`` tests on CPU cover it.
"""
import torch
import torch.distributed as dist


def shard_graphs(num_graphs, rank, world_size):
    """graph ids of this rank's shard: contiguous, sizes differ by at most one."""
    base, rem = divmod(num_graphs, world_size)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def broadcast_parameters(modules, src=0):
    """make every rank start from rank ``src``'s weights and buffers (one flat broadcast)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    tensors = [t for m in modules for t in list(m.parameters()) + list(m.buffers())]
    for dtype in sorted({t.dtype for t in tensors}, key=str):  # identical order on every rank
        group = [t.data for t in tensors if t.dtype == dtype]
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src)
        off = 0
        for t in group:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


class GradBucket:
    """One flat buffer for all gradients of a parameter list; ``allreduce()`` = pack -> one
    all_reduce(SUM) -> scale -> unpack.  ``weight`` (this rank's share of the global loss
    normaliser, e.g. local_masked/global_masked) reproduces the single-process big-batch gradient
    exactly; the default 1/world_size is the usual DDP mean."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def allreduce(self, weight=None):
        if not dist.is_initialized():
            return
        world = dist.get_world_size()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self.views, grads)
        self.flat.mul_(weight if weight is not None else 1.0 / world)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        for p, g in zip(self.params, grads):
            if p.grad is None:
                p.grad = g
        torch._foreach_copy_(grads, self.views)  # one multi-tensor kernel back into the .grad tensors


class AllReduceOptimizers:
    """Wraps the optimizer list of the reference ``train()`` (chem/pretrain_masking.py:134-138) so
    that the unchanged loop body -- zero_grad x3, backward, step x3 -- becomes data parallel: the
    first ``step()`` after a backward all-reduces the one flat gradient bucket, then every wrapped
    optimizer applies its update."""

    def __init__(self, optimizers, weight_fn=None):
        params = [p for o in optimizers for g in o.param_groups for p in g["params"]]
        self.bucket = GradBucket(params)
        self._pending = True
        self.weight_fn = weight_fn
        self.optimizers = [_Wrapped(o, self) for o in optimizers]

    def __iter__(self):
        return iter(self.optimizers)

    def __getitem__(self, i):
        return self.optimizers[i]

    def __len__(self):
        return len(self.optimizers)

    def _before_step(self):
        if self._pending:
            self.bucket.allreduce(self.weight_fn() if self.weight_fn else None)
            self._pending = False


class _Wrapped:
    def __init__(self, opt, owner):
        self.opt, self.owner = opt, owner

    def zero_grad(self, *a, **k):
        self.owner._pending = True
        return self.opt.zero_grad(*a, **k)

    def step(self, *a, **k):
        self.owner._before_step()
        return self.opt.step(*a, **k)

    def __getattr__(self, name):
        return getattr(self.opt, name)


def init_from_env(backend=None):
    """torchrun-style bootstrap: one process per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from env."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("PGNN_DP_FORCE_INIT") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # "nccl" is RCCL on ROCm; PGNN_DP_BACKEND=gloo is a debugging aid
            backend = os.environ.get("PGNN_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            # one process per GPU; more ranks than GPUs (several ranks sharing a device) only makes sense
            # for functional tests with gloo -- RCCL refuses duplicate devices
            local = local % torch.cuda.device_count() if backend != "nccl" else local
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
