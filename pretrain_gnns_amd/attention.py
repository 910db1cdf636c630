"""Attention-flavoured parts of the class surface: GATConv message passing, GlobalAttention and
Set2Set pooling.  Not on the north-star hot path (GIN / GCN), but native since round 2:

* the chem and the bio GATConv with the reference's default two heads run on csrc/attention.hip (``ops.GATAggregate`` /
  ``ops.BioGATAggregate``: CSR edge soft-max + weighted aggregate, deterministic; for bio the ``Linear(9, 2D)`` edge term is
  folded into per-node feature sums, the [E, 2D] edge embedding is never formed).  ``gat_propagate`` below, a composition of
  torch GPU ops like the reference's own torch_geometric path, is what both classes fall back to for ``heads != 2`` only;
* GlobalAttention / Set2Set take their soft-max from ``ops.segment_softmax`` (pgnn_segment_softmax_*) and their
  weighted sums from the deterministic segment-sum kernel (``ops.global_add_pool``).
"""
import torch
import torch.nn.functional as F

from . import ops


def segment_softmax(src, index, num_segments):
    """torch_geometric.utils.softmax (1.0.3): per segment subtract the max, exp, divide by sum + 1e-16.  The pinned
    torch_scatter 1.1.2 pre-fills scatter_max's output with 0, i.e. the shift is max(0, segment max) -- reproduced.
    (torch-op form, used by ``gat_propagate``, the heads != 2 fallback, only.)"""
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    mx = torch.zeros((num_segments,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    den = torch.zeros((num_segments,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device).index_add_(0, index, out)
    return out / (den[index] + 1e-16)


def glorot_(tensor):
    a = (6.0 / (tensor.size(-2) + tensor.size(-1))) ** 0.5
    return tensor.data.uniform_(-a, a)


def gat_propagate(xh, edge_index, edge_emb, self_emb, att, bias, heads, negative_slope):
    """message / softmax / aggregate / update of GATConv (chem/model.py:147-162, bio/model.py:165-180).

    xh [N, heads*D] projected node features; edge_emb [E, heads*D] per-edge embedding; self_emb
    [heads*D] the self-loop embedding; returns [N, D]."""
    n, d = xh.size(0), xh.size(1) // heads
    loop = torch.arange(n, device=xh.device, dtype=edge_index.dtype)
    dst = torch.cat([edge_index[0], loop])
    src = torch.cat([edge_index[1], loop])
    ee = torch.cat([edge_emb, self_emb.unsqueeze(0).expand(n, -1)], dim=0).view(-1, heads, d)
    xh = xh.view(n, heads, d)
    x_j = xh[src] + ee
    alpha = (xh[dst] * att[:, :, :d]).sum(-1) + (x_j * att[:, :, d:]).sum(-1)
    alpha = segment_softmax(F.leaky_relu(alpha, negative_slope), dst, n)
    out = torch.zeros(n, heads, d, dtype=xh.dtype, device=xh.device).index_add_(0, dst, x_j * alpha.unsqueeze(-1))
    return out.mean(dim=1) + bias


class GlobalAttention(torch.nn.Module):
    """torch_geometric.nn.GlobalAttention(gate_nn) as used by chem/model.py:329-333: soft-max of the gate over the nodes
    of a graph (HIP segment soft-max), then the gate-weighted sum (HIP segment sum)."""

    def __init__(self, gate_nn):
        super().__init__()
        self.gate_nn = gate_nn

    def forward(self, x, batch, size=None):
        size = int(batch.max().item()) + 1 if size is None else size
        gate = ops.segment_softmax(self.gate_nn(x).view(-1, 1), batch, size)
        return ops.global_add_pool(gate * x, batch, size)


class Set2Set(torch.nn.Module):
    """torch_geometric.nn.Set2Set(in_channels, processing_steps) as used by chem/model.py:334-339."""

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
            a = ops.segment_softmax((x * q[batch]).sum(dim=-1, keepdim=True), batch, size)
            r = ops.global_add_pool(a * x, batch, size)
            q_star = torch.cat([q, r], dim=-1)
        return q_star
