"""Restatement of the three torch_geometric==1.0.3 / torch_scatter==1.1.2 primitives
the reference hot path calls.  Test infrastructure only (see oracle/__init__.py).

The wheels are pinned at /root/reference/requirements.txt:4-5 and are not vendored, so
their published behaviour is restated here:

* ``add_self_loops(edge_index, num_nodes)`` (called at chem/model.py:39, bio/model.py:39):
  in 1.0.3 it returns ONE tensor, ``cat([edge_index, [[0..N-1],[0..N-1]]], dim=1)`` --
  self loops are appended AFTER the real edges.
* ``MessagePassing.propagate(aggr, edge_index, **kw)`` (chem/model.py:49,101):
  arguments whose name ends in ``_j`` are gathered with ``edge_index[1]``, ``_i`` with
  ``edge_index[0]``; the message is reduced with ``scatter_(aggr, msg, edge_index[0],
  dim_size=N)``; then ``update``.
* ``torch_scatter.scatter_add(src, index, dim=0, dim_size=N)`` == a zero tensor that
  receives ``scatter_add_`` -- on CPU a sequential loop in index order.
* ``global_mean_pool(x, batch)`` (chem/model.py:326, chem/pretrain_contextpred.py:32):
  ``size = batch.max()+1``; sum per graph divided by ``count.clamp(min=1)``.
"""
import torch


def add_self_loops(edge_index, num_nodes):
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], dim=1)


def scatter_add(src, index, dim_size):
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def propagate_add(edge_index, x_j_source, per_edge_term, combine, num_nodes):
    """aggregate at edge_index[0] the message combine(x[edge_index[1]], per_edge_term)."""
    msg = combine(x_j_source[edge_index[1]], per_edge_term)
    return scatter_add(msg, edge_index[0], num_nodes)


def scatter_mean(src, index, dim_size):
    """torch_scatter 1.1.2 scatter_mean: the sum divided by the per-index count clamped to >= 1."""
    total = scatter_add(src, index, dim_size)
    count = scatter_add(torch.ones(src.size(0), dtype=src.dtype, device=src.device), index, dim_size)
    return total / count.clamp(min=1).unsqueeze(-1)


def propagate_mean(edge_index, x_j_source, per_edge_term, combine, num_nodes):
    """aggr='mean' of MessagePassing.propagate (GraphSAGEConv, chem/model.py:167,196)."""
    msg = combine(x_j_source[edge_index[1]], per_edge_term)
    return scatter_mean(msg, edge_index[0], num_nodes)


def softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax of 1.0.3 (chem/model.py:155): per index group, subtract the group
    max, exponentiate, divide by the group sum + 1e-16.  The max comes from torch_scatter 1.1.2's
    scatter_max, whose output is pre-filled with fill_value = 0: the shift is max(0, group max), so a group
    whose scores are all very negative is NOT rescued from underflow (the 1e-16 then dominates)."""
    shape = (num_nodes,) + tuple(src.shape[1:])
    mx = torch.zeros(shape, dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, index.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    return out / (scatter_add(out, index, num_nodes)[index] + 1e-16)


def glorot_(tensor):
    """torch_geometric.nn.inits.glorot: U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))."""
    a = (6.0 / (tensor.size(-2) + tensor.size(-1))) ** 0.5
    return tensor.data.uniform_(-a, a)


class GlobalAttention(torch.nn.Module):
    """torch_geometric.nn.GlobalAttention(gate_nn) of 1.0.3 (chem/model.py:329-333): softmax of the gate
    over the nodes of a graph, then the gate-weighted sum."""

    def __init__(self, gate_nn):
        super().__init__()
        self.gate_nn = gate_nn

    def forward(self, x, batch, size=None):
        size = int(batch.max().item()) + 1 if size is None else size
        gate = softmax(self.gate_nn(x).view(-1, 1), batch, size)
        return scatter_add(gate * x, batch, size)


class Set2Set(torch.nn.Module):
    """torch_geometric.nn.Set2Set(in_channels, processing_steps) of 1.0.3 (chem/model.py:334-339)."""

    def __init__(self, in_channels, processing_steps, num_layers=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, 2 * in_channels
        self.processing_steps, self.num_layers = processing_steps, num_layers
        self.lstm = torch.nn.LSTM(self.out_channels, self.in_channels, num_layers)
        self.lstm.reset_parameters()  # 1.0.3's Set2Set.reset_parameters(): a second draw, kept for seeded-init parity

    def forward(self, x, batch):
        size = int(batch.max().item()) + 1
        h = (x.new_zeros((self.num_layers, size, self.in_channels)), x.new_zeros((self.num_layers, size, self.in_channels)))
        q_star = x.new_zeros(size, self.out_channels)
        for _ in range(self.processing_steps):
            q, h = self.lstm(q_star.unsqueeze(0), h)
            q = q.view(size, self.in_channels)
            e = (x * q[batch]).sum(dim=-1, keepdim=True)
            a = softmax(e, batch, size)
            r = scatter_add(a * x, batch, size)
            q_star = torch.cat([q, r], dim=-1)
        return q_star


def global_add_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    return scatter_add(x, batch, size)


def global_mean_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    total = scatter_add(x, batch, size)
    count = scatter_add(torch.ones(x.size(0), dtype=x.dtype, device=x.device), batch, size)
    return total / count.clamp(min=1).unsqueeze(-1)


def global_max_pool(x, batch, size=None):
    """scatter_('max') of torch_geometric 1.0.3: filled with -1e38, the fill value replaced by 0 afterwards (a graph
    without nodes pools to 0)."""
    size = int(batch.max().item()) + 1 if size is None else size
    out = torch.full((size, x.size(1)), -1e38, dtype=x.dtype, device=x.device)
    out = out.scatter_reduce(0, batch.unsqueeze(-1).expand_as(x), x, reduce="amax", include_self=True)
    return torch.where(out == -1e38, torch.zeros_like(out), out)
