"""Data parallelism for the pre-training hot path: shard by graph, one collective per step.

The reference is single-device (chem/pretrain_masking.py:114); this layer is new design, not a
translation.  A batch is a block-diagonal union of independent graphs (chem/batch.py:31-52), so
the path shards with no data-path exchange: every rank (one process per GPU) collates its own
graphs, runs forward/backward locally, and the only collective is ONE sum-all-reduce per step of
a single flat fp32 bucket holding every gradient (GNN 7.43 MB + heads; RCCL over xGMI picks
direct reduce-scatter/all-gather on the fully connected 8-GPU node), followed by identical Adam
updates on every rank.  BatchNorm statistics stay per rank by default (standard DDP semantics);
``use_exact_batchnorm(model)`` switches the outer BatchNorm layers to statistics all-reduced over the ranks
(sum x, sum x^2: two more tiny collectives per layer and direction).  With shared statistics one rank's rows carry gradient
terms of EVERY rank's loss (the backward all-reduces sum dy and sum dy.xhat), so the rank's share of the global loss
normaliser -- local_M / global_M -- must multiply its LOSS before ``backward()`` and the bucket must plainly sum
(``weight_fn=lambda: 1.0``): that makes the N-rank step EXACTLY the single-process step on the global batch.  Scaling the
finished gradients instead (``weight_fn = local_M / global_M``) is exact only with per-rank statistics.

Aliasing contract: after the first ``allreduce()`` every ``p.grad`` IS a view of the flat bucket -- holding or detaching a
``.grad`` aliases bucket storage, ``zero_grad(set_to_none=False)`` gradients are scaled in place by the next pack, and
anything that writes the bucket between ``backward()`` and ``optimizer.step()`` destroys live gradients.

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


def top_of_network(gnn, from_layer):
    """the parameters of a chem / bio ``GNN`` whose gradients a one-call backward has finished once layer ``from_layer`` is
    enqueued: ``gnns[l]`` and ``batch_norms[l]`` for l >= from_layer (the backward runs from the top layer down; the atom
    embeddings come last)"""
    out = []
    for l in range(from_layer, len(gnn.gnns)):
        out += list(gnn.gnns[l].parameters()) + list(gnn.batch_norms[l].parameters())
    return out


class GradBucket:
    """One flat buffer for all gradients of a parameter list; ``allreduce()`` = pack -> scale -> one
    all_reduce(SUM) (RCCL: one all_reduce(AVG)) -> ``p.grad`` = the bucket's views.  ``weight`` (this rank's share of the global loss
    normaliser, e.g. local_masked/global_masked) reproduces the single-process big-batch gradient
    exactly; the default 1/world_size is the usual DDP mean.

    ``late`` (optional): parameters whose gradients arrive LAST in a backward -- they are laid out behind everything else, so
    that ``allreduce_overlapped`` can reduce the head of the buffer on a communication stream while the backward is still
    producing the tail (see ``AllReduceOptimizers(overlap=...)``)."""

    def __init__(self, params, late=()):
        late_ids = {id(p) for p in late}
        params = [p for p in params if p.requires_grad]
        self.params = [p for p in params if id(p) not in late_ids] + [p for p in params if id(p) in late_ids]
        self.n_early = sum(1 for p in params if id(p) not in late_ids)
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for i, p in enumerate(self.params):
            if i == self.n_early:
                self.split = off
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if self.n_early == len(self.params):
            self.split = off

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def _pack(self, lo, hi, stream=None):
        """gradients of params[lo:hi] into their views; returns False if one of them already lived in the bucket (it got there by
        an in-place accumulation on the caller's stream, i.e. late)"""
        src, dst, fresh = [], [], True
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            if g.data_ptr() != v.data_ptr():  # (a gradient kept from the last step already lives in the bucket)
                src.append(g)
                dst.append(v)
                if stream is not None and g.is_cuda:
                    g.record_stream(stream)
            else:
                fresh = False
        if dst:
            torch._foreach_copy_(dst, src)
        return fresh

    def _reduce(self, flat, weight, async_op=False):
        if weight is None and dist.get_backend() == "nccl":
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=async_op)  # RCCL scales each contribution by 1/world itself
        flat.mul_(weight if weight is not None else 1.0 / dist.get_world_size())
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)

    def _adopt(self):
        for p, v in zip(self.params, self.views):
            p.grad = v  # the reduced gradients stay where they are: the optimizers read the bucket, nothing is copied back

    def allreduce(self, weight=None):
        if not dist.is_initialized():
            return
        self._pack(0, len(self.params))
        self._reduce(self.flat, weight)
        self._adopt()

    def allreduce_overlapped(self, weight, comm_stream, wait_for_milestone):
        """the same result as ``allreduce`` in two collectives: the head of the buffer (everything but ``late``) on
        ``comm_stream`` as soon as ``wait_for_milestone(comm_stream)`` says its gradients are behind events the stream now
        waits for, the tail on the caller's stream behind the whole backward.  Falls back to ordering ``comm_stream`` behind
        the caller's stream when there is no milestone, or when a head gradient reached the bucket by in-place accumulation."""
        if not dist.is_initialized():
            return
        cur = torch.cuda.current_stream(self.flat.device)
        in_place = any(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                       for p, v in zip(self.params[:self.n_early], self.views[:self.n_early]))
        if in_place or not wait_for_milestone(comm_stream):
            comm_stream.wait_stream(cur)
        with torch.cuda.stream(comm_stream):
            self._pack(0, self.n_early, stream=comm_stream)
            work = self._reduce(self.flat[:self.split], weight, async_op=True)
        if self.split < self.flat.numel():
            self._pack(self.n_early, len(self.params))
            self._reduce(self.flat[self.split:], weight)
        if work is not None:
            work.wait()
        cur.wait_stream(comm_stream)
        self._adopt()


class AllReduceOptimizers:
    """Wraps the optimizer list of the reference ``train()`` (chem/pretrain_masking.py:134-138) so
    that the unchanged loop body -- zero_grad x3, backward, step x3 -- becomes data parallel: the
    first ``step()`` after a backward all-reduces the one flat gradient bucket, then every wrapped
    optimizer applies its update."""

    def __init__(self, optimizers, weight_fn=None, overlap=None):
        """``overlap=(gnn, from_layer[, other_networks])`` (opt-in; chem GIN one-call network with direct gradient deposit, CUDA; any
        further one-call network whose parameters these optimizers hold must be listed: its gradients are reduced last): the all-reduce of
        everything the backward has finished once layer ``from_layer`` of ``gnn`` is enqueued -- the heads, layers >=
        from_layer -- runs on a communication stream under the backward of the layers below; the rest (those layers, the atom
        embeddings) follows on the caller's stream.  Same sums, same bits as the single collective."""
        params = [p for o in optimizers for g in o.param_groups for p in g["params"]]
        self.overlap_layer, self.comm_stream, late = None, None, ()
        self._overlap_w1, self._autograd_backwards = None, 0
        if overlap is not None and params and params[0].is_cuda:
            gnn, from_layer = overlap[0], overlap[1]
            others = overlap[2] if len(overlap) > 2 else ()
            top = {id(p) for p in top_of_network(gnn, from_layer)}
            late = [p for p in gnn.parameters() if id(p) not in top]  # layers below from_layer, the atom embeddings
            # every OTHER one-call network under these optimizers (context prediction: the context network) records no milestone
            # -- the arming is keyed on `gnn` -- so none of its gradients is behind the events: they go last too
            for net in others:
                late += list(net.parameters())
            self.overlap_layer = int(from_layer)
            self._overlap_w1 = gnn.gnns[from_layer].mlp[0].weight  # pgnn_gin_layer.w1 of that layer: the network's identity
            self.comm_stream = torch.cuda.Stream(device=params[0].device)
        self.bucket = GradBucket(params, late=late)
        self._pending = True
        self.weight_fn = weight_fn
        self.overlapped_steps = 0  # steps whose head collective waited for the backward's milestone, not for the whole backward
        self.optimizers = [_Wrapped(o, self) for o in optimizers]
        self._arm()

    def __iter__(self):
        return iter(self.optimizers)

    def __getitem__(self, i):
        return self.optimizers[i]

    def __len__(self):
        return len(self.optimizers)

    def _arm(self):
        if self.overlap_layer is not None and dist.is_initialized():
            from . import _lib, ops
            self._autograd_backwards = ops.autograd_path_backwards()
            _lib.load().pgnn_stack_bwd_milestone_arm(self.overlap_layer, self._overlap_w1.data_ptr())

    def _wait_for_milestone(self, stream):
        from . import _lib, ops
        # autograd's AccumulateGrad writes .grad behind the whole backward: with direct deposit off, or when ANY stack backward since
        # the arming fell back to it (hooks on a parameter), the events cover nothing
        if not ops.direct_grads_enabled() or ops.autograd_path_backwards() != self._autograd_backwards:
            return False
        # (the C side refuses when no backward of the armed network, or more than one -- gradient accumulation -- reached the layer)
        ok = _lib.load().pgnn_stack_bwd_milestone_wait(stream.cuda_stream) == 0
        self.overlapped_steps += int(ok)
        return ok

    def _before_step(self):
        if self._pending:
            weight = self.weight_fn() if self.weight_fn else None
            if weight is not None and not isinstance(weight, (int, float)):
                # a device tensor produced on the caller's stream would be read on the communication stream unordered
                raise TypeError("weight_fn must return a Python float (fetch a device scalar with .item() first), got %s" % type(weight).__name__)
            if self.overlap_layer is not None:
                self.bucket.allreduce_overlapped(weight, self.comm_stream, self._wait_for_milestone)
            else:
                self.bucket.allreduce(weight)
            self._pending = False


class _Wrapped:
    def __init__(self, opt, owner):
        self.opt, self.owner = opt, owner

    def zero_grad(self, *a, **k):
        if not self.owner._pending:
            self.owner._arm()  # once per step: the next backward records its gradient milestone
        self.owner._pending = True
        return self.opt.zero_grad(*a, **k)

    def step(self, *a, **k):
        self.owner._before_step()
        return self.opt.step(*a, **k)

    def __getattr__(self, name):
        return getattr(self.opt, name)


class _ExactBN(torch.autograd.Function):
    """y = relu?((x - mean) * invstd * gamma + beta) with mean / biased variance over the rows of ALL ranks."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, stats_out):
        n_local = x.size(0)
        xd = x.double()
        packed = torch.cat([xd.sum(0), (xd * xd).sum(0), xd.new_tensor([float(n_local)])])
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        d = x.size(1)
        n = float(packed[-1].item())
        mean = packed[:d] / n
        var = (packed[d:2 * d] / n - mean * mean).clamp_(min=0.0)
        invstd = (var + eps).rsqrt()
        xhat = ((xd - mean) * invstd).to(x.dtype)
        y = xhat * gamma + beta
        if relu:
            y = torch.relu(y)
        stats_out.append((mean.to(x.dtype), var.to(x.dtype), n))
        ctx.save_for_backward(xhat, gamma, invstd.to(x.dtype), y if relu else x.new_empty(0))
        ctx.relu, ctx.n = bool(relu), n
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, gamma, invstd, y = ctx.saved_tensors
        if ctx.relu:
            dy = dy * (y > 0).to(dy.dtype)
        dyd, xh = dy.double(), xhat.double()
        s_dy, s_dyx = dyd.sum(0), (dyd * xh).sum(0)
        dgamma, dbeta = s_dyx.to(dy.dtype), s_dy.to(dy.dtype)  # local parts: the gradient bucket sums them over ranks
        packed = torch.cat([s_dy, s_dyx])
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        d = dy.size(1)
        g_dy, g_dyx = packed[:d] / ctx.n, packed[d:] / ctx.n
        dx = ((dyd - g_dy - xh * g_dyx) * (gamma.double() * invstd.double())).to(dy.dtype)
        return dx, dgamma, dbeta, None, None, None


class ExactBatchNorm1d(torch.nn.BatchNorm1d):
    """BatchNorm1d whose training-mode statistics span the rows of every data-parallel rank (SURVEY 8e "exact
    mode").  Same parameters, buffers and state-dict keys as ``torch.nn.BatchNorm1d``; eval mode is the ordinary
    running-statistics form.  Off the tuned path: the reductions are torch GPU ops and the GNN takes its per-layer
    route (no one-call network) while such modules are installed.

    Loss weighting: with shared statistics the backward of one rank's rows carries terms of every rank's loss, so a
    per-rank loss share (local_M / global_M for the masked-atom mean) must multiply the LOSS before ``backward()``
    and the bucket must then plainly sum (``AllReduceOptimizers(..., weight_fn=lambda: 1.0)``); scaling finished
    gradients (``weight_fn`` = the share) is exact only with per-rank statistics."""

    pgnn_exact = True

    def forward(self, x, relu=False):
        if not (self.training or self.running_mean is None):
            y = torch.nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)
            return torch.relu(y) if relu else y
        stats = []
        y = _ExactBN.apply(x, self.weight, self.bias, self.eps, relu, stats)
        if self.track_running_stats and self.running_mean is not None:
            mean, var, n = stats[0]
            self.num_batches_tracked.add_(1)
            m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
            with torch.no_grad():
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(var * (n / max(n - 1.0, 1.0)), alpha=m)
        return y


def use_exact_batchnorm(module):
    """replace every ``torch.nn.BatchNorm1d`` below ``module`` by an ``ExactBatchNorm1d`` that shares its parameter
    and buffer tensors (optimizers built before or after keep working; state dicts are unchanged)"""
    for name, child in list(module.named_children()):
        if type(child) is torch.nn.BatchNorm1d:
            new = ExactBatchNorm1d(child.num_features, eps=child.eps, momentum=child.momentum, affine=child.affine,
                                   track_running_stats=child.track_running_stats)
            new._parameters, new._buffers, new.training = child._parameters, child._buffers, child.training
            setattr(module, name, new)
        else:
            use_exact_batchnorm(child)
    return module


def comm_report(optimizers, iters=10):
    """what the collective of a step costs here: backend, world size, bucket size and the HIP-event time of one
    flat-bucket all-reduce (bench.py prints this so that a scaling run shows RCCL really saw N ranks)"""
    if not dist.is_initialized():
        return {"initialized": False, "world": 1}
    bucket = optimizers.bucket if isinstance(optimizers, AllReduceOptimizers) else None
    out = {"initialized": True, "backend": dist.get_backend(), "world": dist.get_world_size(), "rank": dist.get_rank(),
           "bucket_bytes": bucket.nbytes if bucket is not None else None}
    if bucket is not None and bucket.flat.is_cuda:
        scratch = torch.zeros_like(bucket.flat)  # same size, NOT the bucket: its gradients may be live
        for _ in range(3):
            dist.all_reduce(scratch)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            dist.all_reduce(scratch)
        e.record()
        torch.cuda.synchronize()
        out["allreduce_us"] = round(s.elapsed_time(e) / iters * 1e3, 1)
        if out["world"] > 1:  # ring / direct all-reduce moves 2 (W-1)/W of the bucket per rank
            out["allreduce_busbw_GBps"] = round(2.0 * (out["world"] - 1) / out["world"] * bucket.nbytes / (out["allreduce_us"] * 1e-6) / 1e9, 1)
    return out


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
