"""torch.autograd bindings of the HIP hot path (include/pgnn.h) -- host-side plumbing only.

Every function here launches hand-written gfx950 kernels through the C ABI on
``torch.cuda.current_stream()``; PyTorch supplies device memory, streams and the autograd tape,
nothing else.  Inputs must live on the GPU: there is deliberately no CPU / eager fallback.
"""
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import check, load, require_cuda, stream_ptr

_CHECK_INDICES = os.environ.get("PGNN_CHECK_INDICES", "0") == "1"
_NO_ATTACHED_GRAPH = os.environ.get("PGNN_LOADER_STRUCTURE", "1") == "0"  # A/B: always build the structure from the COO
_BIO_TILES = os.environ.get("PGNN_BIO_TILES", "1") != "0"  # A/B: graph-resident bio aggregation (csrc/tile.hip)
_ws_cache = {}


def _ws_bytes(fn_name, *shape):
    key = (fn_name,) + shape
    v = _ws_cache.get(key)
    if v is None:
        v = int(getattr(load(), fn_name)(*shape))
        _ws_cache[key] = v
    return v


_ws_pool = {}


def _workspace(nbytes, device):
    """scratch for ONE C call.  Every library call finishes with its workspace before it returns
    control to the stream's next kernel (all work is enqueued in order on the current stream), so a
    single grow-only buffer per (device, stream) replaces one allocator round trip per op."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, stream_ptr(idx))
    buf = _ws_pool.get(key)
    need = max(int(nbytes), 256)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1 << 20) * 2 if buf is not None else max(need, 1 << 22), dtype=torch.uint8,
                          device=device)
        _ws_pool[key] = buf
    return buf


def _f32c(t):
    if t.dtype != torch.float32:
        raise _lib.PgnnError("expected float32 tensor, got %s" % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def _rows2d(t):
    """accept a 2-D fp32 tensor whose rows are contiguous (row stride = leading dimension)."""
    if t.dtype != torch.float32:
        raise _lib.PgnnError("expected float32 tensor, got %s" % t.dtype)
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 4 != 0 or t.data_ptr() % 16 != 0:
        t = t.contiguous()
    return t


# ------------------------------------------------------------------------------------ graph
class GraphStruct:
    """CSR-by-destination + CSR-by-source + per-node edge-feature sums of one batch."""

    __slots__ = ("kind", "gcn", "n", "e", "in_ptr", "in_src", "in_code", "out_ptr", "out_dst", "dinv", "cfeat",
                 "status", "tiles", "slot_feat")

    @property
    def kc(self):
        return self.cfeat.size(1)

    def check(self):
        bad = int(self.status.item())
        if bad:
            raise IndexError("%d out-of-range node indices / edge attributes in the batch" % bad)


def _build_graph(kind, edge_index, edge_attr, num_nodes, gcn):
    require_cuda(edge_index, edge_attr)
    lib = load()
    dev = edge_index.device
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise _lib.PgnnError("edge_index must be int64 [2, E]")
    ei = edge_index.contiguous()
    e, n = ei.size(1), int(num_nodes)
    g = GraphStruct()
    g.kind, g.gcn, g.n, g.e, g.tiles, g.slot_feat = kind, bool(gcn), n, e, None, None
    g.in_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    g.out_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    g.in_src = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    g.out_dst = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    g.dinv = torch.empty(n, dtype=torch.float32, device=dev)
    g.status = _lib.status_word(dev)
    ws = _workspace(_ws_bytes("pgnn_graph_workspace_bytes", n, e), dev)
    if kind == "chem":
        if edge_attr.dtype != torch.int64 or edge_attr.dim() != 2 or edge_attr.size(1) != 2:
            raise _lib.PgnnError("chem edge_attr must be int64 [E, 2]")
        ea = edge_attr.contiguous()
        g.in_code = torch.empty(max(e, 1), dtype=torch.uint8, device=dev)
        g.cfeat = torch.empty(n, 9, dtype=torch.float32, device=dev)
        check(lib.pgnn_chem_graph_build(ei.data_ptr(), ea.data_ptr(), e, n, int(g.gcn), g.in_ptr.data_ptr(),
                                        g.in_src.data_ptr(), g.in_code.data_ptr(), g.out_ptr.data_ptr(),
                                        g.out_dst.data_ptr(), g.dinv.data_ptr(), g.cfeat.data_ptr(),
                                        g.status.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()),
              "pgnn_chem_graph_build")
    else:
        if edge_attr.dim() != 2 or edge_attr.size(1) != 9:
            raise _lib.PgnnError("bio edge_attr must be [E, 9]")
        ea = edge_attr.to(torch.float32).contiguous()
        g.in_code = None
        g.cfeat = torch.empty(n, 10, dtype=torch.float32, device=dev)
        check(lib.pgnn_bio_graph_build(ei.data_ptr(), ea.data_ptr(), e, n, int(g.gcn), g.in_ptr.data_ptr(),
                                       g.in_src.data_ptr(), g.out_ptr.data_ptr(), g.out_dst.data_ptr(),
                                       g.dinv.data_ptr(), g.cfeat.data_ptr(), g.status.data_ptr(), ws.data_ptr(),
                                       ws.numel(), stream_ptr()),
              "pgnn_bio_graph_build")
        if _BIO_TILES:  # closed node intervals (= the ego nets) for the graph-resident aggregation of csrc/tile.hip
            tile_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
            num_tiles = torch.empty(1, dtype=torch.int32, device=dev)
            tws = _workspace(_ws_bytes("pgnn_graph_tiles_workspace_bytes", n), dev)
            check(lib.pgnn_graph_tiles(g.in_ptr.data_ptr(), g.in_src.data_ptr(), g.out_ptr.data_ptr(), g.out_dst.data_ptr(), n,
                                       tile_start.data_ptr(), num_tiles.data_ptr(), tws.data_ptr(), tws.numel(), stream_ptr()),
                  "pgnn_graph_tiles")
            g.tiles = (tile_start, num_tiles)
    if _CHECK_INDICES:
        g.check()
    return g


def attach_graph(edge_index, edge_attr, graph):
    """The loader's hand-over of a batch's structure (data/resident.py: built by offset-add from the dataset's):
    it rides on the ``edge_index`` tensor OBJECT, valid for as long as that tensor and ``edge_attr`` are the ones the loader made and
    nobody has written to them (``Tensor._version``).  The class surface is untouched -- ``GNN.forward(x, edge_index, edge_attr)``
    looks here first (``attached_graph``) and builds from the COO whenever anything differs."""
    import weakref
    edge_index._pgnn_graph = (graph, edge_index._version, weakref.ref(edge_attr), edge_attr._version)


def attached_graph(kind, edge_index, edge_attr, num_nodes, gcn):
    rec = getattr(edge_index, "_pgnn_graph", None)
    if rec is None or _NO_ATTACHED_GRAPH:
        return None
    g, v_ei, ea_ref, v_ea = rec
    if (ea_ref() is not edge_attr or edge_index._version != v_ei or edge_attr._version != v_ea or g.kind != kind or g.gcn != bool(gcn)
            or g.n != int(num_nodes) or g.e != edge_index.size(1) or g.in_ptr.device != edge_index.device):
        return None
    return g


def build_chem_graph(edge_index, edge_attr, num_nodes, gcn=False, reuse=False):
    """``reuse``: take the structure the loader attached to this very ``edge_index`` if there is one (the model's forward)"""
    if reuse:
        g = attached_graph("chem", edge_index, edge_attr, num_nodes, gcn)
        if g is not None:
            return g
    return _build_graph("chem", edge_index, edge_attr, num_nodes, gcn)


def build_bio_graph(edge_index, edge_attr, num_nodes, gcn=False):
    return _build_graph("bio", edge_index, edge_attr, num_nodes, gcn)


def group_by_key(key, n_keys, stride=1, offset=0):
    """stable grouping of items by int64 key -> (ptr[n_keys+1], perm[n_items]) int32."""
    require_cuda(key)
    dev = key.device
    n_items = key.numel() // stride
    ptr = torch.empty(n_keys + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(n_items, 1), dtype=torch.int32, device=dev)
    status = _lib.status_word(dev)
    ws = _workspace(_ws_bytes("pgnn_group_workspace_bytes", n_keys, n_items), dev)
    check(load().pgnn_group_by_key(key.data_ptr() + 8 * offset, stride, n_items, n_keys, ptr.data_ptr(),
                                   perm.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr()),
          "pgnn_group_by_key")
    if _CHECK_INDICES and int(status.item()):
        raise IndexError("key out of range in group_by_key")
    return ptr, perm


# ------------------------------------------------------------------------------------ raw launches
def _neighbor_sum(x, ptr, nbr, dinv, n, dim, out=None, tiles=None, feat=None):
    """out = neighbour sum of x (+ self).  ``tiles``: the graph-resident kernel of csrc/tile.hip; ``feat`` =
    (cfeat, table, feat_out) folds ``feat_out (+)= cfeat . table`` into the same launch (tiles only)."""
    x = _rows2d(x)
    if out is None:
        out = torch.empty(n, dim, dtype=torch.float32, device=x.device)
    if tiles is not None:
        cf, tb, fo = feat if feat is not None else (None, None, None)
        check(load().pgnn_neighbor_sum_tiled(x.data_ptr(), x.stride(0), ptr.data_ptr(), nbr.data_ptr(),
                                             dinv.data_ptr() if dinv is not None else None, tiles[0].data_ptr(),
                                             tiles[1].data_ptr(), out.data_ptr(), out.stride(0), n, dim,
                                             cf.data_ptr() if cf is not None else None, cf.size(1) if cf is not None else 0,
                                             tb.data_ptr() if tb is not None else None, tb.stride(0) if tb is not None else 0,
                                             fo.data_ptr() if fo is not None else None, fo.stride(0) if fo is not None else 0,
                                             stream_ptr()), "pgnn_neighbor_sum_tiled")
        return out
    assert feat is None
    check(load().pgnn_neighbor_sum(x.data_ptr(), x.stride(0), ptr.data_ptr(), nbr.data_ptr(),
                                   dinv.data_ptr() if dinv is not None else None, out.data_ptr(), out.stride(0),
                                   n, dim, stream_ptr()), "pgnn_neighbor_sum")
    return out


def _rowfeat_bwd(cfeat, g, dim):
    g = _rows2d(g)
    n, kc = cfeat.size(0), cfeat.size(1)
    gt = torch.empty(kc, dim, dtype=torch.float32, device=g.device)
    ws = _workspace(_ws_bytes("pgnn_rowfeat_matmul_bwd_workspace_bytes", n, kc, dim), g.device)
    check(load().pgnn_rowfeat_matmul_bwd(cfeat.data_ptr(), kc, g.data_ptr(), g.stride(0), gt.data_ptr(), dim, n, dim,
                                         ws.data_ptr(), ws.numel(), stream_ptr()), "pgnn_rowfeat_matmul_bwd")
    return gt


def _rowfeat_fwd(cfeat, table, out, dim, accumulate):
    table = _f32c(table)
    check(load().pgnn_rowfeat_matmul_fwd(cfeat.data_ptr(), cfeat.size(1), table.data_ptr(), table.stride(0),
                                         out.data_ptr(), out.stride(0), cfeat.size(0), dim, int(accumulate),
                                         stream_ptr()), "pgnn_rowfeat_matmul_fwd")


# ------------------------------------------------------------------------------------ aggregation
class ChemAggregate(Function):
    """chem GINConv / GCNConv message + scatter_add (chem/model.py:47-52, 95-104)."""

    @staticmethod
    def forward(ctx, x, emb1, emb2, graph):
        require_cuda(x, emb1, emb2)
        x, emb1, emb2 = _rows2d(x), _f32c(emb1), _f32c(emb2)
        n, dim = x.shape
        if n != graph.n or emb1.shape != (6, dim) or emb2.shape != (3, dim):
            raise _lib.PgnnError("chem aggregate: shape mismatch")
        out = torch.empty(n, dim, dtype=torch.float32, device=x.device)
        check(load().pgnn_chem_aggregate_fwd(x.data_ptr(), x.stride(0), graph.in_ptr.data_ptr(),
                                             graph.in_src.data_ptr(), graph.in_code.data_ptr(), emb1.data_ptr(),
                                             emb2.data_ptr(), graph.dinv.data_ptr() if graph.gcn else None,
                                             out.data_ptr(), out.stride(0), n, dim, stream_ptr()),
              "pgnn_chem_aggregate_fwd")
        ctx.graph = graph
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        graph = ctx.graph
        g = _rows2d(g)
        n, dim = g.shape
        gx = ge1 = ge2 = None
        if ctx.needs_input_grad[0]:
            gx = _neighbor_sum(g, graph.out_ptr, graph.out_dst, graph.dinv if graph.gcn else None, n, dim)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gt = _rowfeat_bwd(graph.cfeat, g, dim)
            ge1, ge2 = gt[:6], gt[6:9]
        return gx, ge1, ge2, None


class BioAggregate(Function):
    """bio GINConv (concat message, bio/model.py:47,52-55) / GCNConv (bio/model.py:98-114) aggregation.

    GIN: out[N,2D] = [sum_e x_j + x_i , cfeat . [W^T; b]]; GCN: out[N,D] = sum_e n_e x_j + cfeat . [W^T; b].
    """

    @staticmethod
    def forward(ctx, x, enc_w, enc_b, graph):
        require_cuda(x, enc_w, enc_b)
        x = _rows2d(x)
        n, dim = x.shape
        if n != graph.n or enc_w.shape != (dim, 9):
            raise _lib.PgnnError("bio aggregate: shape mismatch")
        table = torch.cat([enc_w.t(), enc_b.unsqueeze(0)], dim=0).contiguous()  # [10, D]
        if graph.gcn:
            if graph.tiles is not None:  # neighbour sum and edge-feature product in ONE graph-resident launch
                out = torch.empty(n, dim, dtype=torch.float32, device=x.device)
                _neighbor_sum(x, graph.in_ptr, graph.in_src, graph.dinv, n, dim, out=out, tiles=graph.tiles,
                              feat=(graph.cfeat, table, out))
            else:
                out = _neighbor_sum(x, graph.in_ptr, graph.in_src, graph.dinv, n, dim)
                _rowfeat_fwd(graph.cfeat, table, out, dim, True)
        else:
            out = torch.empty(n, 2 * dim, dtype=torch.float32, device=x.device)
            if graph.tiles is not None:
                _neighbor_sum(x, graph.in_ptr, graph.in_src, None, n, dim, out=out[:, :dim], tiles=graph.tiles,
                              feat=(graph.cfeat, table, out[:, dim:]))
            else:
                _neighbor_sum(x, graph.in_ptr, graph.in_src, None, n, dim, out=out[:, :dim])
                _rowfeat_fwd(graph.cfeat, table, out[:, dim:], dim, False)
        ctx.graph, ctx.dim = graph, dim
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        graph, dim = ctx.graph, ctx.dim
        g = _rows2d(g)
        n = g.size(0)
        gx_part, ge_part = (g, g) if graph.gcn else (g[:, :dim], g[:, dim:])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _neighbor_sum(gx_part, graph.out_ptr, graph.out_dst, graph.dinv if graph.gcn else None, n, dim, tiles=graph.tiles)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gt = _rowfeat_bwd(graph.cfeat, ge_part, dim)
            gw, gb = gt[:9].t(), gt[9]
        return gx, gw, gb, None


class BioSumAggregate(Function):
    """bio GraphSAGEConv message + unweighted sum (bio/model.py:200-221): out[N,D] = sum_e (x_j + enc(e_ij))
    incl. the self loop; ``graph`` must be built with gcn=False (plain per-node edge-feature sums)."""

    @staticmethod
    def forward(ctx, x, enc_w, enc_b, graph):
        require_cuda(x, enc_w, enc_b)
        x = _rows2d(x)
        n, dim = x.shape
        if n != graph.n or enc_w.shape != (dim, 9) or graph.gcn:
            raise _lib.PgnnError("bio sum aggregate: shape mismatch / graph built with GCN weights")
        table = torch.cat([enc_w.t(), enc_b.unsqueeze(0)], dim=0).contiguous()  # [10, D]
        if graph.tiles is not None:
            out = torch.empty(n, dim, dtype=torch.float32, device=x.device)
            _neighbor_sum(x, graph.in_ptr, graph.in_src, None, n, dim, out=out, tiles=graph.tiles, feat=(graph.cfeat, table, out))
        else:
            out = _neighbor_sum(x, graph.in_ptr, graph.in_src, None, n, dim)
            _rowfeat_fwd(graph.cfeat, table, out, dim, True)
        ctx.graph, ctx.dim = graph, dim
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        graph, dim = ctx.graph, ctx.dim
        g = _rows2d(g)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _neighbor_sum(g, graph.out_ptr, graph.out_dst, None, g.size(0), dim, tiles=graph.tiles)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gt = _rowfeat_bwd(graph.cfeat, g, dim)
            gw, gb = gt[:9].t(), gt[9]
        return gx, gw, gb, None


class MeanL2Normalize(Function):
    """GraphSAGE update on top of an unweighted aggregation: divide by (in-degree + 1), then
    F.normalize(p=2, dim=-1)  (chem/model.py:167,201-202; bio/model.py:183,223-224)."""

    @staticmethod
    def forward(ctx, total, graph):
        require_cuda(total)
        total = _rows2d(total)
        n, dim = total.shape
        y = torch.empty(n, dim, dtype=torch.float32, device=total.device)
        norm = torch.empty(n, dtype=torch.float32, device=total.device)
        check(load().pgnn_mean_l2norm_fwd(total.data_ptr(), total.stride(0), graph.in_ptr.data_ptr(), y.data_ptr(), dim,
                                          norm.data_ptr(), n, dim, stream_ptr()), "pgnn_mean_l2norm_fwd")
        ctx.save_for_backward(y, norm)
        ctx.graph = graph
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        y, norm = ctx.saved_tensors
        dy = _rows2d(dy)
        n, dim = y.shape
        d = torch.empty(n, dim, dtype=torch.float32, device=y.device)
        check(load().pgnn_mean_l2norm_bwd(dy.data_ptr(), dy.stride(0), y.data_ptr(), dim, norm.data_ptr(),
                                          ctx.graph.in_ptr.data_ptr(), d.data_ptr(), dim, n, dim, stream_ptr()),
              "pgnn_mean_l2norm_bwd")
        return d, None


# ------------------------------------------------------------------------------------ embedding
class Embed(Function):
    """x_embedding1(x[:,0]) + x_embedding2(x[:,1]) (chem/model.py:264) or a single table
    (bio/model.py:49-50).  ``idx`` is int64 [N, ncol]."""

    @staticmethod
    def forward(ctx, idx, table1, table2):
        require_cuda(idx, table1, table2)
        if idx.dtype != torch.int64:
            raise _lib.PgnnError("embedding indices must be int64")
        idx = idx.contiguous()
        n = idx.size(0)
        stride = idx.size(1) if idx.dim() == 2 else 1
        t1 = _f32c(table1)
        t2 = _f32c(table2) if table2 is not None else None
        dim = t1.size(1)
        out = torch.empty(n, dim, dtype=torch.float32, device=idx.device)
        status = _lib.status_word(idx.device)
        check(load().pgnn_embed_fwd(idx.data_ptr(), stride, t1.data_ptr(), t1.size(0),
                                    t2.data_ptr() if t2 is not None else None, t2.size(0) if t2 is not None else 0,
                                    out.data_ptr(), dim, n, dim, status.data_ptr(), stream_ptr()), "pgnn_embed_fwd")
        if _CHECK_INDICES and int(status.item()):
            raise IndexError("embedding index out of range")
        ctx.idx, ctx.stride, ctx.rows = idx, stride, (t1.size(0), t2.size(0) if t2 is not None else 0)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _rows2d(g)
        idx, stride = ctx.idx, ctx.stride
        grads = []
        for col, rows in enumerate(ctx.rows):
            if rows == 0 or not ctx.needs_input_grad[1 + col]:
                grads.append(None)
                continue
            ptr, perm = group_by_key(idx, rows, stride=stride, offset=col)
            grads.append(segment_sum_raw(g, ptr, perm, idx.size(0), rows, mean=False))
        return None, grads[0], grads[1]


def segment_sum_raw(x, ptr, perm, n_items, n_seg, mean):
    x = _rows2d(x)
    dim = x.size(1)
    out = torch.empty(n_seg, dim, dtype=torch.float32, device=x.device)
    ws = _workspace(_ws_bytes("pgnn_segment_sum_workspace_bytes", n_items, n_seg, dim), x.device)
    check(load().pgnn_segment_sum(x.data_ptr(), x.stride(0), ptr.data_ptr(), perm.data_ptr() if perm is not None else None,
                                  n_items, n_seg, int(mean), out.data_ptr(), dim, dim, ws.data_ptr(), ws.numel(),
                                  stream_ptr()), "pgnn_segment_sum")
    return out


class SegmentPool(Function):
    """global_add_pool / global_mean_pool (chem/model.py:324-326; torch_geometric 1.0.3 scatter_)."""

    @staticmethod
    def forward(ctx, x, batch, size, mean):
        require_cuda(x, batch)
        x = _rows2d(x)
        batch = batch.contiguous()
        ptr, perm = group_by_key(batch, size)
        out = segment_sum_raw(x, ptr, perm, x.size(0), size, mean)
        ctx.batch, ctx.ptr, ctx.mean = batch, ptr, mean
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _rows2d(g)
        n, dim = ctx.batch.numel(), g.size(1)
        gx = torch.empty(n, dim, dtype=torch.float32, device=g.device)
        check(load().pgnn_segment_broadcast(g.data_ptr(), g.stride(0), ctx.batch.data_ptr(), ctx.ptr.data_ptr(),
                                            int(ctx.mean), gx.data_ptr(), dim, n, dim, stream_ptr()),
              "pgnn_segment_broadcast")
        return gx, None, None, None


def global_add_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    return SegmentPool.apply(x, batch, size, False)


def global_mean_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    return SegmentPool.apply(x, batch, size, True)


# ------------------------------------------------------------------------------------ attention layers (csrc/attention.hip)
class GATAggregate(Function):
    """message / edge soft-max / aggregate / update of the 2-head chem GATConv (chem/model.py:133-162) on the CSR of
    the graph build: xh [N, 2D] = weight_linear(x) -> out [N, D].  Parameters: att [1, 2, 2D], bias [D], emb1 [6, 2D],
    emb2 [3, 2D].  All sums sequential in a fixed order: bitwise reproducible (the torch-op composition it replaces
    used atomic index_add_)."""

    @staticmethod
    def forward(ctx, xh, att, bias, emb1, emb2, graph, negative_slope):
        require_cuda(xh, att, bias, emb1, emb2)
        xh = _rows2d(xh)
        n, hd = xh.shape
        heads, d = 2, hd // 2
        if att.shape != (1, heads, 2 * d) or emb1.shape != (6, hd) or emb2.shape != (3, hd) or graph.kind != "chem" or n != graph.n:
            raise _lib.PgnnError("GAT: shape mismatch (2 heads, chem graph)")
        dev = xh.device
        att2 = _f32c(att.view(heads, 2 * d))
        e1, e2, b = _f32c(emb1), _f32c(emb2), _f32c(bias)
        # parameter-space precomputation: the bond term of the logits, ctab[c, h] = (emb1[c // 3] + emb2[c % 3])[h] . att_j[h]
        table = (e1.view(6, 1, heads, d) + e2.view(1, 3, heads, d)).view(18, heads, d)
        ctab = (table * att2[:, d:].unsqueeze(0)).sum(-1).contiguous()
        slots = graph.e + n
        scores = torch.empty(n, 2 * heads, dtype=torch.float32, device=dev)
        z = torch.empty(slots, heads, dtype=torch.float32, device=dev)
        alpha = torch.empty(slots, heads, dtype=torch.float32, device=dev)
        cfa = torch.empty(heads, n, 9, dtype=torch.float32, device=dev)
        out = torch.empty(n, d, dtype=torch.float32, device=dev)
        check(load().pgnn_gat_fwd(xh.data_ptr(), xh.stride(0), graph.in_ptr.data_ptr(), graph.in_src.data_ptr(),
                                  graph.in_code.data_ptr(), e1.data_ptr(), e2.data_ptr(), ctab.data_ptr(), None, None, None, 0,
                                  att2.data_ptr(), b.data_ptr(), float(negative_slope), scores.data_ptr(), z.data_ptr(),
                                  alpha.data_ptr(), cfa.data_ptr(), out.data_ptr(), d, n, d, stream_ptr()), "pgnn_gat_fwd")
        ctx.save_for_backward(xh, att2, e1, e2, z, alpha, cfa)
        ctx.graph, ctx.slope = graph, float(negative_slope)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xh, att2, e1, e2, z, alpha, cfa = ctx.saved_tensors
        graph = ctx.graph
        g = _rows2d(g)
        n, hd = xh.shape
        heads, d = 2, hd // 2
        dev = g.device
        dalpha = torch.empty_like(alpha)
        dsd = torch.empty(heads, n, 2, dtype=torch.float32, device=dev)
        czf = torch.empty(heads, n, 9, dtype=torch.float32, device=dev)
        wout = torch.empty(max(graph.e, 1), heads, dtype=torch.float32, device=dev)
        dxh = torch.empty(n, hd, dtype=torch.float32, device=dev)
        check(load().pgnn_gat_bwd(g.data_ptr(), g.stride(0), xh.data_ptr(), xh.stride(0), graph.in_ptr.data_ptr(),
                                  graph.in_src.data_ptr(), graph.in_code.data_ptr(), graph.out_ptr.data_ptr(),
                                  graph.out_dst.data_ptr(), e1.data_ptr(), e2.data_ptr(), None, None, None, 0, att2.data_ptr(),
                                  ctx.slope, z.data_ptr(), alpha.data_ptr(), dalpha.data_ptr(), dsd.data_ptr(), czf.data_ptr(),
                                  wout.data_ptr(), dxh.data_ptr(), hd, n, d, stream_ptr()), "pgnn_gat_bwd")
        demb = torch.empty(9, hd, dtype=torch.float32, device=dev)  # rows 0..5 = d emb1, 6..8 = d emb2
        datt = torch.empty(heads, 2 * d, dtype=torch.float32, device=dev)
        e9 = torch.cat([e1, e2], dim=0)
        lib, sp = load(), stream_ptr()
        for h in range(heads):
            cols = slice(h * d, (h + 1) * d)
            ws = _workspace(_ws_bytes("pgnn_rowfeat_matmul_bwd_workspace_bytes", n, 9, d), dev)
            # message path: dE[t] = sum_i cfa[h, i, t] * g[i]   (cfa already carries the 1/heads of the head mean)
            check(lib.pgnn_rowfeat_matmul_bwd(cfa[h].data_ptr(), 9, g.data_ptr(), g.stride(0), demb[:, cols].data_ptr(), hd, n, d,
                                              ws.data_ptr(), ws.numel(), sp), "pgnn_rowfeat_matmul_bwd")
            # logits path through the bond term: sum of dz per type / direction
            s9 = czf[h].sum(0)
            demb[:, cols] += s9.unsqueeze(1) * att2[h, d:].unsqueeze(0)
            # d att: [dst term; src term] = dsd[h]^T . xh[:, head h]
            xh_h = xh[:, cols]
            gt = torch.empty(2, d, dtype=torch.float32, device=dev)
            check(lib.pgnn_rowfeat_matmul_bwd(dsd[h].data_ptr(), 2, xh_h.data_ptr(), xh.stride(0), gt.data_ptr(), d, n, d,
                                              ws.data_ptr(), ws.numel(), sp), "pgnn_rowfeat_matmul_bwd")
            datt[h, :d] = gt[0]
            datt[h, d:] = gt[1] + (s9.unsqueeze(1) * e9[:, cols]).sum(0)
        return dxh, datt.view(1, heads, 2 * d), g.sum(0), demb[:6], demb[6:], None, None


def bio_slot_features(graph, edge_index, edge_attr):
    """[E, 10] fp32: the 9 edge attributes in CSR-slot order (the stable grouping by edge_index[0] the graph build uses)
    followed by a constant 1 -- the bias column of edge_encoder.  Built once per batch and shared by the layers."""
    cached = getattr(graph, "slot_feat", None)
    if cached is None:
        e = graph.e
        _, perm = group_by_key(edge_index[0].contiguous(), graph.n)
        attrs = edge_attr.to(torch.float32)[perm[:e].long()]
        cached = torch.cat([attrs, torch.ones(e, 1, dtype=torch.float32, device=attrs.device)], dim=1).contiguous()
        if e == 0:
            cached = torch.zeros(1, 10, dtype=torch.float32, device=attrs.device)
        graph.slot_feat = cached
    return cached


class BioGATAggregate(Function):
    """message / edge soft-max / aggregate / update of the 2-head bio GATConv (bio/model.py:147-180): xh [N, 2D] =
    weight_linear(x) -> out [N, D], edge term edge_encoder(attr) = W_enc attr + b_enc (self loops: attr = one-hot 7).
    The [E, 2D] encoder output is never formed (linearity; csrc/attention.hip header).  Sums sequential in a fixed order."""

    KF = 10

    @staticmethod
    def forward(ctx, xh, att, bias, enc_w, enc_b, graph, slot_feat, negative_slope):
        require_cuda(xh, att, bias, enc_w, enc_b, slot_feat)
        xh = _rows2d(xh)
        n, hd = xh.shape
        heads, d, kf = 2, hd // 2, BioGATAggregate.KF
        if att.shape != (1, heads, 2 * d) or enc_w.shape != (hd, kf - 1) or enc_b.shape != (hd,) or graph.kind != "bio" \
                or n != graph.n or slot_feat.shape != (max(graph.e, 1), kf):
            raise _lib.PgnnError("bio GAT: shape mismatch (2 heads, bio graph, 9 edge attributes)")
        dev = xh.device
        att2 = _f32c(att.view(heads, 2 * d))
        b = _f32c(bias)
        tenc = torch.cat([enc_w.t(), enc_b.unsqueeze(0)], dim=0).to(torch.float32).contiguous()  # [kf, 2D]
        wv = (tenc.view(kf, heads, d) * att2[:, d:].unsqueeze(0)).sum(-1).contiguous()  # [kf, heads]
        self_feat = torch.zeros(kf, dtype=torch.float32, device=dev)
        self_feat[7] = 1.0
        self_feat[kf - 1] = 1.0
        slots = graph.e + n
        scores = torch.empty(n, 2 * heads, dtype=torch.float32, device=dev)
        z = torch.empty(slots, heads, dtype=torch.float32, device=dev)
        alpha = torch.empty(slots, heads, dtype=torch.float32, device=dev)
        cfa = torch.empty(heads, n, kf, dtype=torch.float32, device=dev)
        out = torch.empty(n, d, dtype=torch.float32, device=dev)
        check(load().pgnn_gat_fwd(xh.data_ptr(), xh.stride(0), graph.in_ptr.data_ptr(), graph.in_src.data_ptr(), None, None, None,
                                  None, slot_feat.data_ptr(), self_feat.data_ptr(), wv.data_ptr(), kf, att2.data_ptr(),
                                  b.data_ptr(), float(negative_slope), scores.data_ptr(), z.data_ptr(), alpha.data_ptr(),
                                  cfa.data_ptr(), out.data_ptr(), d, n, d, stream_ptr()), "pgnn_gat_fwd")
        for h in range(heads):  # + (sum_e a_eh f_e / heads) . Tenc_h
            _rowfeat_fwd(cfa[h], tenc[:, h * d:(h + 1) * d], out, d, accumulate=True)
        ctx.save_for_backward(xh, att2, tenc, slot_feat, self_feat, z, alpha, cfa)
        ctx.graph, ctx.slope = graph, float(negative_slope)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xh, att2, tenc, slot_feat, self_feat, z, alpha, cfa = ctx.saved_tensors
        graph = ctx.graph
        g = _rows2d(g)
        n, hd = xh.shape
        heads, d, kf = 2, hd // 2, BioGATAggregate.KF
        dev = g.device
        dalpha = torch.empty_like(alpha)
        dsd = torch.empty(heads, n, 2, dtype=torch.float32, device=dev)
        czf = torch.empty(heads, n, kf, dtype=torch.float32, device=dev)
        wout = torch.empty(max(graph.e, 1), heads, dtype=torch.float32, device=dev)
        dxh = torch.empty(n, hd, dtype=torch.float32, device=dev)
        check(load().pgnn_gat_bwd(g.data_ptr(), g.stride(0), xh.data_ptr(), xh.stride(0), graph.in_ptr.data_ptr(),
                                  graph.in_src.data_ptr(), None, graph.out_ptr.data_ptr(), graph.out_dst.data_ptr(), None, None,
                                  slot_feat.data_ptr(), self_feat.data_ptr(), tenc.data_ptr(), kf, att2.data_ptr(), ctx.slope,
                                  z.data_ptr(), alpha.data_ptr(), dalpha.data_ptr(), dsd.data_ptr(), czf.data_ptr(),
                                  wout.data_ptr(), dxh.data_ptr(), hd, n, d, stream_ptr()), "pgnn_gat_bwd")
        dtenc = torch.empty(kf, hd, dtype=torch.float32, device=dev)
        datt = torch.empty(heads, 2 * d, dtype=torch.float32, device=dev)
        lib, sp = load(), stream_ptr()
        for h in range(heads):
            cols = slice(h * d, (h + 1) * d)
            ws = _workspace(_ws_bytes("pgnn_rowfeat_matmul_bwd_workspace_bytes", n, kf, d), dev)
            check(lib.pgnn_rowfeat_matmul_bwd(cfa[h].data_ptr(), kf, g.data_ptr(), g.stride(0), dtenc[:, cols].data_ptr(), hd, n, d,
                                              ws.data_ptr(), ws.numel(), sp), "pgnn_rowfeat_matmul_bwd")
            sk = czf[h].sum(0)  # logits path through the edge term
            dtenc[:, cols] += sk.unsqueeze(1) * att2[h, d:].unsqueeze(0)
            gt = torch.empty(2, d, dtype=torch.float32, device=dev)
            check(lib.pgnn_rowfeat_matmul_bwd(dsd[h].data_ptr(), 2, xh[:, cols].data_ptr(), xh.stride(0), gt.data_ptr(), d, n, d,
                                              ws.data_ptr(), ws.numel(), sp), "pgnn_rowfeat_matmul_bwd")
            datt[h, :d] = gt[0]
            datt[h, d:] = gt[1] + (sk.unsqueeze(1) * tenc[:, cols]).sum(0)
        return dxh, datt.view(1, heads, 2 * d), g.sum(0), dtenc[:kf - 1].t(), dtenc[kf - 1], None, None, None


def _segments(batch, size):
    """(ptr, perm) of the items grouped by their int64 segment key (group_by_key: stable, any order of `batch`)"""
    return group_by_key(batch.contiguous(), size)


class SegmentSoftmax(Function):
    """torch_geometric.utils.softmax (1.0.3) over the segments given by (ptr, perm); z [items] or [items, heads]."""

    @staticmethod
    def forward(ctx, z, ptr, perm, size):
        require_cuda(z)
        shape = z.shape
        z2 = _f32c(z.reshape(shape[0], -1))
        alpha = torch.empty_like(z2)
        check(load().pgnn_segment_softmax_fwd(z2.data_ptr(), ptr.data_ptr(), perm.data_ptr(), alpha.data_ptr(), size, z2.size(1),
                                              stream_ptr()), "pgnn_segment_softmax_fwd")
        ctx.save_for_backward(alpha)
        ctx.ptr, ctx.perm, ctx.size, ctx.shape = ptr, perm, size, shape
        return alpha.view(shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        g2 = _f32c(g.reshape(alpha.shape))
        dz = torch.empty_like(alpha)
        check(load().pgnn_segment_softmax_bwd(alpha.data_ptr(), g2.data_ptr(), ctx.ptr.data_ptr(), ctx.perm.data_ptr(),
                                              dz.data_ptr(), ctx.size, alpha.size(1), stream_ptr()), "pgnn_segment_softmax_bwd")
        return dz.view(ctx.shape), None, None, None


def segment_softmax(z, batch, size):
    ptr, perm = _segments(batch, size)
    return SegmentSoftmax.apply(z, ptr, perm, size)


class SegmentMax(Function):
    """global_max_pool (chem/model.py:327-328; torch_geometric 1.0.3 scatter_('max'): empty graph -> 0)."""

    @staticmethod
    def forward(ctx, x, batch, size):
        require_cuda(x, batch)
        x = _rows2d(x)
        n, dim = x.shape
        if dim % 4:
            raise _lib.PgnnError("global_max_pool: feature width must be a multiple of 4")
        batch = batch.contiguous()
        ptr, perm = _segments(batch, size)
        out = torch.empty(size, dim, dtype=torch.float32, device=x.device)
        arg = torch.empty(size, dim, dtype=torch.int32, device=x.device)
        check(load().pgnn_segment_max_fwd(x.data_ptr(), x.stride(0), ptr.data_ptr(), perm.data_ptr(), out.data_ptr(), dim,
                                          arg.data_ptr(), size, dim, stream_ptr()), "pgnn_segment_max_fwd")
        ctx.batch, ctx.arg, ctx.n = batch, arg, n
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _rows2d(g)
        size, dim = g.shape
        dx = torch.empty(ctx.n, dim, dtype=torch.float32, device=g.device)
        check(load().pgnn_segment_max_bwd(g.data_ptr(), g.stride(0), ctx.batch.data_ptr(), ctx.arg.data_ptr(), dx.data_ptr(), dim,
                                          size, ctx.n, dim, stream_ptr()), "pgnn_segment_max_bwd")
        return dx, None, None


def global_max_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    return SegmentMax.apply(x, batch, size)


# ------------------------------------------------------------------------------------ batch norm
class BatchNormReLU(Function):
    """BatchNorm1d (+ optional fused ReLU); chem/model.py:269-275, bio/model.py:24."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, drop_p=0.0,
                drop_seed=0):
        require_cuda(x, gamma, beta)
        x = _rows2d(x)
        n, dim = x.shape
        if training and n <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        y = torch.empty(n, dim, dtype=torch.float32, device=x.device)
        save_mean = torch.empty(dim, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(dim, dtype=torch.float32, device=x.device)
        ws = _workspace(_ws_bytes("pgnn_bn_workspace_bytes", n, dim), x.device)
        check(load().pgnn_bn_fwd(x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                 running_mean.data_ptr() if running_mean is not None else None,
                                 running_var.data_ptr() if running_var is not None else None, float(momentum),
                                 float(eps), int(training), int(relu), y.data_ptr(), dim, save_mean.data_ptr(),
                                 save_invstd.data_ptr(), float(drop_p), int(drop_seed), n, dim, ws.data_ptr(),
                                 ws.numel(), stream_ptr()),
              "pgnn_bn_fwd")
        ctx.save_for_backward(x, gamma, beta, save_mean, save_invstd)
        ctx.training, ctx.relu, ctx.drop = bool(training), bool(relu), (float(drop_p), int(drop_seed))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, gamma, beta, save_mean, save_invstd = ctx.saved_tensors
        dy = _rows2d(dy)
        n, dim = x.shape
        dx = torch.empty(n, dim, dtype=torch.float32, device=x.device)
        dgamma = torch.empty(dim, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(dim, dtype=torch.float32, device=x.device)
        ws = _workspace(_ws_bytes("pgnn_bn_workspace_bytes", n, dim), x.device)
        check(load().pgnn_bn_bwd(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), gamma.data_ptr(),
                                 beta.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(), int(ctx.training),
                                 int(ctx.relu), dx.data_ptr(), dim, dgamma.data_ptr(), dbeta.data_ptr(), ctx.drop[0],
                                 ctx.drop[1], n, dim, ws.data_ptr(), ws.numel(), stream_ptr()), "pgnn_bn_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


def dropout_seed():
    """64-bit seed for one fused-dropout call, drawn from torch's CPU generator (so torch.manual_seed
    makes runs reproducible) without touching the device."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def batch_norm(x, bn, relu, drop_p=0.0):
    """apply a torch.nn.BatchNorm1d module's parameters/buffers with the HIP kernels; ``drop_p`` > 0
    fuses the inverted dropout that follows it in GNN.forward."""
    if getattr(bn, "pgnn_exact", False):  # parallel.ExactBatchNorm1d: statistics all-reduced over the DP ranks
        y = bn(x, relu=relu)
        return torch.nn.functional.dropout(y, drop_p, training=True) if drop_p > 0 else y
    training = bn.training or bn.running_mean is None
    momentum = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            momentum = 1.0 / float(bn.num_batches_tracked)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    if not bn.affine:
        raise _lib.PgnnError("BatchNorm1d without affine parameters is not on the hot path")
    return BatchNormReLU.apply(x, bn.weight, bn.bias, rm, rv, training, momentum, bn.eps, relu, drop_p,
                               dropout_seed() if drop_p > 0 else 0)


# ------------------------------------------------------------------------------------ linear layers
def _linear_fwd(x, w, b, relu):
    m, k = x.shape
    n = w.size(0)
    y = torch.empty(m, n, dtype=torch.float32, device=x.device)
    check(load().pgnn_linear_fwd(x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr() if b is not None else None,
                                 y.data_ptr(), n, m, k, n, int(relu), stream_ptr()), "pgnn_linear_fwd")
    return y


def _linear_bwd_data(dy, w, relu_out):
    m, n = dy.shape
    k = w.size(1)
    dx = torch.empty(m, k, dtype=torch.float32, device=dy.device)
    check(load().pgnn_linear_bwd_data(dy.data_ptr(), dy.stride(0), w.data_ptr(),
                                      relu_out.data_ptr() if relu_out is not None else None,
                                      relu_out.stride(0) if relu_out is not None else 0, dx.data_ptr(), k, m, k, n,
                                      stream_ptr()), "pgnn_linear_bwd_data")
    return dx


def _linear_bwd_weight(dy, x, need_bias):
    m, n = dy.shape
    k = x.size(1)
    dw = torch.empty(n, k, dtype=torch.float32, device=dy.device)
    db = torch.empty(n, dtype=torch.float32, device=dy.device) if need_bias else None
    ws = _workspace(_ws_bytes("pgnn_linear_bwd_weight_workspace_bytes", m, k, n), dy.device)
    check(load().pgnn_linear_bwd_weight(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(),
                                        db.data_ptr() if db is not None else None, m, k, n, ws.data_ptr(), ws.numel(),
                                        stream_ptr()), "pgnn_linear_bwd_weight")
    return dw, db


class Linear(Function):
    """y = x W^T + b (GCN linear chem/model.py:99; bio mlp layers bio/model.py:24)."""

    @staticmethod
    def forward(ctx, x, w, b):
        require_cuda(x, w, b)
        x, w = _rows2d(x), _f32c(w)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return _linear_fwd(x, w, _f32c(b) if b is not None else None, False)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _rows2d(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _linear_bwd_data(dy, w, None)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = _linear_bwd_weight(dy, x, ctx.has_bias)
        return dx, dw, db


class MLP2(Function):
    """GIN update: Linear -> ReLU -> Linear (chem/model.py:29,54-55) with the hidden activation kept
    once and the ReLU mask fused into the backward-data GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        require_cuda(x, w1, b1, w2, b2)
        x, w1, w2 = _rows2d(x), _f32c(w1), _f32c(w2)
        hid = _linear_fwd(x, w1, _f32c(b1), True)
        out = _linear_fwd(hid, w2, _f32c(b2), False)
        ctx.save_for_backward(x, w1, w2, hid)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, w1, w2, hid = ctx.saved_tensors
        dout = _rows2d(dout)
        dw2, db2 = _linear_bwd_weight(dout, hid, True)
        dhid = _linear_bwd_data(dout, w2, hid)  # masked by hid > 0
        dw1, db1 = _linear_bwd_weight(dhid, x, True)
        dx = _linear_bwd_data(dhid, w1, None) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2


def linear(x, layer):
    return Linear.apply(x, layer.weight, layer.bias)


# ------------------------------------------------------------------------------------ masking prediction head
_head_words = {}


def _head_state(dev):
    """[status, -, arrival tickets (PGNN_TICKET_WORDS)]: int32 words per (device, stream), zeroed once (the kernels leave the
    tickets at zero).  Per stream like the workspace: two heads launched concurrently on two streams of one device (a side stream,
    two ranks in threads on one GPU) must not count their blocks in the same words."""
    if torch.cuda.is_current_stream_capturing():  # a captured launch writes its words on every replay: its own, never cached or evicted
        return torch.zeros(64, dtype=torch.int32, device=dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, stream_ptr(idx))
    t = _head_words.pop(key, None)
    if t is None:
        if len(_head_words) >= 16:  # streams come and go: the least recently USED entry goes (dicts keep insertion order;
            _head_words.pop(next(iter(_head_words)))  # a hit re-inserts its key at the end)
        t = torch.zeros(64, dtype=torch.int32, device=dev)
    _head_words[key] = t
    return t


import threading as _threading

_row_support_tls = _threading.local()  # .rec = (data_ptr of a gradient tensor, its rows, the int64 rows outside which it is zero):
                                       # MaskedHead.backward -> stack backward, per host thread (autograd runs a device's backward on one thread)


def _hand_over_row_support(dy, n):
    """called by a one-call network's backward with its incoming gradient: if it is the tensor the masking head just produced, tell the
    library which rows are not zero (consumed by the pgnn_*_stack_bwd call that follows)"""
    rec = getattr(_row_support_tls, "rec", None)
    _row_support_tls.rec = None
    if rec is not None and rec[0] == dy.data_ptr() and rec[1] == n and dy.is_contiguous():
        load().pgnn_stack_bwd_dy_rows(dy.data_ptr(), rec[2].data_ptr(), rec[2].numel())
        return rec[2]  # (kept alive by the caller until the launch is enqueued)
    return None


class MaskedHead(Function):
    """linear_pred(node_rep[idx]) -> CrossEntropyLoss()(pred.double(), label) and the number of correct arg-maxes, in one
    launch per direction (chem/pretrain_masking.py:52-57).  Returns (loss float64 [], correct int64 [], logits fp32 [m, C],
    metrics float64 [2] = (loss, correct) for a single read-back); only ``loss`` carries a gradient.  ``idx`` must not
    repeat (MaskAtom samples without replacement)."""

    @staticmethod
    def forward(ctx, node_rep, idx, weight, bias, label, accum=None):
        require_cuda(node_rep, idx, weight, label)
        ctx.set_materialize_grads(False)  # three of the four outputs carry no gradient: no zero tensors for them in the backward
        if accum is not None and (accum.dtype != torch.float64 or accum.numel() < 4 or not accum.is_contiguous() or accum.device != node_rep.device):
            raise _lib.PgnnError("masked head: accum must be a contiguous float64 [4] tensor on the device of node_rep")
        h = _rows2d(node_rep)
        n, dim = h.shape
        m, classes = idx.numel(), weight.size(0)
        if idx.dtype != torch.int64 or label.dtype != torch.int64 or label.size(0) != m or weight.size(1) != dim or m == 0:
            raise _lib.PgnnError("masked head: idx / label must be int64 [m], weight [classes, dim]")
        idx = idx.contiguous()
        w = _f32c(weight)
        b = _f32c(bias) if bias is not None else None
        dev = h.device
        logits = torch.empty(m, classes, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float64, device=dev)
        correct = torch.empty((), dtype=torch.int64, device=dev)
        metrics = torch.empty(2, dtype=torch.float64, device=dev)
        words = _head_state(dev)
        ws = torch.empty(_ws_bytes("pgnn_masked_head_workspace_bytes", m, classes, dim), dtype=torch.uint8, device=dev)
        check(load().pgnn_masked_head_fwd(h.data_ptr(), h.stride(0), n, idx.data_ptr(), m, w.data_ptr(),
                                          b.data_ptr() if b is not None else None, label.data_ptr(), label.stride(0), classes, dim,
                                          logits.data_ptr(), loss.data_ptr(), correct.data_ptr(), metrics.data_ptr(),
                                          accum.data_ptr() if accum is not None else None, words.data_ptr(), words.data_ptr() + 8, ws.data_ptr(), ws.numel(), stream_ptr()), "pgnn_masked_head_fwd")
        if _CHECK_INDICES:
            if int(words[0].item()):
                words[0] = 0
                raise IndexError("masked head: row index or label out of range")
            if torch.unique(idx).numel() != m:  # the backward writes d node_rep rows, it does not accumulate them
                raise ValueError("masked head: repeated row index (MaskAtom samples without replacement)")
        ctx.save_for_backward(h, idx, w, label, logits)
        ctx.ws, ctx.has_bias = ws, bias is not None
        ctx.mark_non_differentiable(correct, logits, metrics)
        return loss, correct, logits, metrics

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss, _gc, _gl, _gm):
        if gloss is None:
            return None, None, None, None, None, None
        h, idx, w, label, logits = ctx.saved_tensors
        n, dim = h.shape
        m, classes = logits.shape
        dev = h.device
        gloss = gloss.to(torch.float64).contiguous()
        dnode = torch.empty(n, dim, dtype=torch.float32, device=dev)
        dw = torch.empty(classes, dim, dtype=torch.float32, device=dev)
        db = torch.empty(classes, dtype=torch.float32, device=dev) if ctx.has_bias else None
        check(load().pgnn_masked_head_bwd(h.data_ptr(), h.stride(0), n, idx.data_ptr(), m, w.data_ptr(), label.data_ptr(),
                                          label.stride(0), logits.data_ptr(), gloss.data_ptr(), classes, dim, dnode.data_ptr(), dim,
                                          dw.data_ptr(), db.data_ptr() if db is not None else None, ctx.ws.data_ptr(), ctx.ws.numel(),
                                          stream_ptr()), "pgnn_masked_head_bwd")
        # dnode is zero outside the rows idx: the one-call network's backward, if this very tensor reaches it, sums its top
        # BatchNorm's column sums over those rows only (pgnn_stack_bwd_dy_rows)
        _row_support_tls.rec = (dnode.data_ptr(), n, idx)
        return dnode, None, dw, db, None, None


def masked_head(node_rep, idx, linear, label, with_metrics=False, accum=None):
    """(loss, correct) of ``linear(node_rep[idx])`` against ``label`` -- see MaskedHead; ``with_metrics`` adds the packed
    float64 [2] tensor (loss, correct); ``accum`` (float64 [4], device) receives accum[0] += loss, accum[1] += correct / m,
    accum[3] += 1: the epoch sums of the reference's train(), kept on the device"""
    loss, correct, _, metrics = MaskedHead.apply(node_rep, idx, linear.weight, linear.bias, label, accum)
    return (loss, correct, metrics) if with_metrics else (loss, correct)


class EdgeHead(Function):
    """linear(node_rep[u] + node_rep[v]) -> CrossEntropyLoss()(pred, label) and the number of correct arg-maxes over the masked
    edges (bio/pretrain_masking.py:45-58; chem/pretrain_masking.py:60-66), csrc/edgehead.hip.  ``ends`` int64 [2, m] =
    ``edge_index[:, masked_idx]``; ``label`` int64 [m] or a float [m, classes] matrix whose first row maximum is the label
    (``torch.argmax(mask_edge_label, 1)``).  ``float64``: the loss in float64 (chem's ``pred.double()``) or fp32 (bio).
    Returns (loss [] in that dtype, correct int64 [], logits fp32 [m, classes], metrics float64 [2] = (loss, correct)); only
    ``loss`` carries a gradient."""

    @staticmethod
    def forward(ctx, node_rep, ends, weight, bias, label, float64=False, accum=None, accum_slot=2, accum_step=True):
        require_cuda(node_rep, ends, weight, label)
        ctx.set_materialize_grads(False)
        if accum is not None and (accum.dtype != torch.float64 or accum.numel() < 4 or not accum.is_contiguous() or accum.device != node_rep.device):
            raise _lib.PgnnError("edge head: accum must be a contiguous float64 [4] tensor on the device of node_rep")
        h = _rows2d(node_rep)
        n, dim = h.shape
        classes = weight.size(0)
        if ends.dtype != torch.int64 or ends.dim() != 2 or ends.size(0) != 2 or ends.size(1) == 0 or weight.size(1) != dim:
            raise _lib.PgnnError("edge head: ends must be int64 [2, m] with m > 0, weight [classes, dim]")
        m = ends.size(1)
        ends = ends.contiguous()
        onehot = label.dtype != torch.int64
        if onehot:
            if label.dtype != torch.float32 or label.dim() != 2 or label.size(0) != m or label.size(1) < classes or label.stride(1) != 1:
                raise _lib.PgnnError("edge head: a float label must be [m, >= classes] fp32 with contiguous rows")
        elif label.dim() != 1 or label.size(0) != m:
            raise _lib.PgnnError("edge head: an int64 label must be [m]")
        w = _f32c(weight)
        b = _f32c(bias) if bias is not None else None
        dev = h.device
        logits = torch.empty(m, classes, dtype=torch.float32, device=dev)
        loss64 = torch.empty((), dtype=torch.float64, device=dev)
        loss32 = None if float64 else torch.empty((), dtype=torch.float32, device=dev)
        correct = torch.empty((), dtype=torch.int64, device=dev)
        metrics = torch.empty(2, dtype=torch.float64, device=dev)
        words = _head_state(dev)
        ws = torch.empty(_ws_bytes("pgnn_edge_head_workspace_bytes", n, m, classes, dim), dtype=torch.uint8, device=dev)
        check(load().pgnn_edge_head_fwd(h.data_ptr(), h.stride(0), n, ends.data_ptr(), m, w.data_ptr(), b.data_ptr() if b is not None else None,
                                        None if onehot else label.data_ptr(), 0 if onehot else label.stride(0),
                                        label.data_ptr() if onehot else None, label.stride(0) if onehot else 0, label.size(1) if onehot else 0,
                                        classes, dim, 1 if float64 else 0, logits.data_ptr(), loss64.data_ptr(), loss32.data_ptr() if loss32 is not None else None,
                                        correct.data_ptr(), metrics.data_ptr(), accum.data_ptr() if accum is not None else None, int(accum_slot),
                                        1 if accum_step else 0, words.data_ptr(), words.data_ptr() + 8, ws.data_ptr(), ws.numel(), stream_ptr()),
              "pgnn_edge_head_fwd")
        if _CHECK_INDICES and int(words[0].item()):
            words[0] = 0
            raise IndexError("edge head: end point or label out of range")
        ctx.save_for_backward(h, ends, w, label, logits)
        ctx.ws, ctx.has_bias, ctx.float64 = ws, bias is not None, bool(float64)
        loss = loss64 if float64 else loss32
        ctx.mark_non_differentiable(correct, logits, metrics)
        return loss, correct, logits, metrics

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss, _gc, _gl, _gm):
        if gloss is None:
            return (None,) * 9
        h, ends, w, label, logits = ctx.saved_tensors
        n, dim = h.shape
        m, classes = logits.shape
        dev = h.device
        onehot = label.dtype != torch.int64
        g64 = gloss.dtype == torch.float64
        gloss = gloss.contiguous() if g64 else gloss.to(torch.float32).contiguous()
        dnode = torch.empty(n, dim, dtype=torch.float32, device=dev)
        dw = torch.empty(classes, dim, dtype=torch.float32, device=dev)
        db = torch.empty(classes, dtype=torch.float32, device=dev) if ctx.has_bias else None
        words = _head_state(dev)
        check(load().pgnn_edge_head_bwd(h.data_ptr(), h.stride(0), n, ends.data_ptr(), m, w.data_ptr(),
                                        None if onehot else label.data_ptr(), 0 if onehot else label.stride(0),
                                        label.data_ptr() if onehot else None, label.stride(0) if onehot else 0, label.size(1) if onehot else 0, logits.data_ptr(),
                                        gloss.data_ptr() if g64 else None, None if g64 else gloss.data_ptr(), classes, dim,
                                        1 if ctx.float64 else 0, dnode.data_ptr(), dim, dw.data_ptr(), db.data_ptr() if db is not None else None,
                                        words.data_ptr() + 8, ctx.ws.data_ptr(), ctx.ws.numel(), stream_ptr()), "pgnn_edge_head_bwd")
        return dnode, None, dw, db, None, None, None, None, None


def edge_head(node_rep, ends, linear, label, float64=False, accum=None, accum_slot=2, accum_step=True):
    """(loss, correct, metrics) of ``linear(node_rep[ends[0]] + node_rep[ends[1]])`` against ``label`` -- see EdgeHead"""
    loss, correct, _, metrics = EdgeHead.apply(node_rep, ends, linear.weight, linear.bias, label, float64, accum, accum_slot, accum_step)
    return loss, correct, metrics


# ------------------------------------------------------------------------------------ context-prediction loss
class ContextPredLoss(Function):
    """chem/pretrain_contextpred.py:54-67,86-97 (cbow, mean context pooling) on the two networks' node embeddings, two launches:
    returns (loss float64 [] = loss_pos + neg_samples * loss_neg -- what train() back-propagates, one launch back --,
    vals float64 [4] = (loss_pos, loss_neg, fraction of pred_pos > 0, fraction of pred_neg < 0), no gradient).
    ``center`` / ``overlap`` must not repeat a row."""

    @staticmethod
    def forward(ctx, hs, center, hc, overlap, seg, neg_samples, accum=None):
        require_cuda(hs, center, hc, overlap, seg)
        ctx.set_materialize_grads(False)
        hs, hc = _rows2d(hs), _rows2d(hc)
        dim, B = hs.size(1), center.numel()
        if hc.size(1) != dim or B == 0 or overlap.numel() != seg.numel() or any(t.dtype != torch.int64 for t in (center, overlap, seg)):
            raise _lib.PgnnError("contextpred loss: hs / hc [*, dim] fp32, center [graphs], overlap / seg [n] int64")
        if accum is not None and (accum.dtype != torch.float64 or accum.numel() < 4 or not accum.is_contiguous() or accum.device != hs.device):
            raise _lib.PgnnError("contextpred loss: accum must be a contiguous float64 [4] tensor on the device of the embeddings")
        center, overlap, seg = center.contiguous(), overlap.contiguous(), seg.contiguous()
        dev = hs.device
        vals = torch.empty(4, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float64, device=dev)
        words = _head_state(dev)
        ws = torch.empty(_ws_bytes("pgnn_contextpred_loss_workspace_bytes", B, dim, int(neg_samples)), dtype=torch.uint8, device=dev)
        check(load().pgnn_contextpred_loss_fwd(hs.data_ptr(), hs.stride(0), hs.size(0), center.data_ptr(), hc.data_ptr(), hc.stride(0), hc.size(0),
                                               overlap.data_ptr(), seg.data_ptr(), overlap.numel(), B, dim, int(neg_samples), vals.data_ptr(),
                                               loss.data_ptr(), accum.data_ptr() if accum is not None else None, words.data_ptr(),
                                               words.data_ptr() + 8, ws.data_ptr(), ws.numel(), stream_ptr()), "pgnn_contextpred_loss_fwd")
        if _CHECK_INDICES and int(words[0].item()):
            words[0] = 0
            raise IndexError("contextpred loss: row index out of range")
        ctx.save_for_backward(hs, center, overlap, seg)
        ctx.ws, ctx.n_ctx, ctx.neg = ws, hc.size(0), int(neg_samples)
        ctx.mark_non_differentiable(vals)
        return loss, vals

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss, _gvals):
        if gloss is None:
            return None, None, None, None, None, None, None
        hs, center, overlap, seg = ctx.saved_tensors
        dev, dim, B = hs.device, hs.size(1), center.numel()
        gloss = gloss.to(torch.float64).contiguous()
        dhs = torch.empty(hs.size(0), dim, dtype=torch.float32, device=dev)
        dhc = torch.empty(ctx.n_ctx, dim, dtype=torch.float32, device=dev)
        check(load().pgnn_contextpred_loss_bwd(hs.data_ptr(), hs.stride(0), hs.size(0), center.data_ptr(), ctx.n_ctx, overlap.data_ptr(),
                                               seg.data_ptr(), overlap.numel(), B, dim, ctx.neg, gloss.data_ptr(), dhs.data_ptr(), dim,
                                               dhc.data_ptr(), dim, ctx.ws.data_ptr(), ctx.ws.numel(), stream_ptr()), "pgnn_contextpred_loss_bwd")
        return dhs, None, dhc, None, None, None, None


def contextpred_loss_eligible(hs, hc, center, neg_samples):
    """can ContextPredLoss take these embeddings?  fp32 [*, dim] with dim % 4 == 0 and dim <= 512 (JK = "concat" is 1 800 wide),
    1 <= neg_samples <= min(8, graphs): the kernel pairs graph g with graph (g + k) % B, which is the reference's cycle_index only
    while k <= B (chem/pretrain_contextpred.py:36-39 raises on the shape mismatch beyond) -- the torch path raises as it does."""
    B = center.numel()
    return (hs.dim() == 2 and hc.dim() == 2 and hs.dtype == torch.float32 and hc.dtype == torch.float32 and hs.size(1) == hc.size(1)
            and hs.size(1) % 4 == 0 and hs.size(1) <= 512 and B > 0 and 1 <= int(neg_samples) <= min(8, B))


def contextpred_loss(hs, center, hc, overlap, seg, neg_samples=1, accum=None):
    if not contextpred_loss_eligible(hs, hc, center, neg_samples):
        raise _lib.PgnnError("contextpred loss: needs fp32 [*, dim] embeddings, dim % 4 == 0, dim <= 512, 1 <= neg_samples <= min(8, graphs)")
    return ContextPredLoss.apply(hs, center, hc, overlap, seg, neg_samples, accum)


# ------------------------------------------------------------------------------------ fused chem GIN layer
class ChemGINLayer(Function):
    """One chem GIN layer + its outer BatchNorm (+ReLU) as ONE library call per direction
    (pgnn_chem_gin_layer_fwd / _bwd): chem/model.py:37-55 and :269-275.  Same kernels, same order and
    therefore bit-identical results to ChemAggregate -> MLP2 -> BatchNormReLU; it exists because at the
    reference's batch size a Python/ctypes/allocator round trip per kernel costs more than the kernel."""

    @staticmethod
    def forward(ctx, x, emb1, emb2, w1, b1, w2, b2, gamma, beta, graph, running_mean, running_var, training,
                momentum, eps, relu, drop_p=0.0, drop_seed=0):
        require_cuda(x, emb1, emb2, w1, b1, w2, b2, gamma, beta)
        x = _rows2d(x)
        n, dim = x.shape
        if training and n <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        dev = x.device
        acts = torch.empty(3, n, dim, dtype=torch.float32, device=dev)  # agg, z, y
        agg, z, y = acts[0], acts[1], acts[2]
        hid = torch.empty(n, 2 * dim, dtype=torch.float32, device=dev)
        stats = torch.empty(2, dim, dtype=torch.float32, device=dev)
        ws = _workspace(_ws_bytes("pgnn_chem_gin_layer_workspace_bytes", n, dim), dev)
        emb1, emb2, w1, b1, w2, b2 = (_f32c(t) for t in (emb1, emb2, w1, b1, w2, b2))
        check(load().pgnn_chem_gin_layer_fwd(
            x.data_ptr(), x.stride(0), graph.in_ptr.data_ptr(), graph.in_src.data_ptr(), graph.in_code.data_ptr(),
            emb1.data_ptr(), emb2.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
            gamma.data_ptr(), beta.data_ptr(),
            running_mean.data_ptr() if running_mean is not None else None,
            running_var.data_ptr() if running_var is not None else None, float(momentum), float(eps), int(training),
            int(relu), agg.data_ptr(), hid.data_ptr(), z.data_ptr(), y.data_ptr(), stats[0].data_ptr(),
            stats[1].data_ptr(), float(drop_p), int(drop_seed), n, dim, ws.data_ptr(), ws.numel(), stream_ptr()),
            "pgnn_chem_gin_layer_fwd")
        ctx.save_for_backward(acts, hid, stats, w1, w2, gamma, beta)
        ctx.graph, ctx.training, ctx.relu, ctx.drop = graph, bool(training), bool(relu), (float(drop_p), int(drop_seed))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        acts, hid, stats, w1, w2, gamma, beta = ctx.saved_tensors
        graph = ctx.graph
        dy = _rows2d(dy)
        agg, z = acts[0], acts[1]
        n, dim = agg.shape
        dev = dy.device
        sizes = [9 * dim, 2 * dim * dim, 2 * dim, 2 * dim * dim, dim, dim, dim]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        demb, dw1, db1, dw2, db2, dgamma, dbeta = torch.split(flat, sizes)
        dx = torch.empty(n, dim, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        ws = _workspace(_ws_bytes("pgnn_chem_gin_layer_workspace_bytes", n, dim), dev)
        check(load().pgnn_chem_gin_layer_bwd(
            dy.data_ptr(), dy.stride(0), agg.data_ptr(), hid.data_ptr(), z.data_ptr(), graph.out_ptr.data_ptr(),
            graph.out_dst.data_ptr(), graph.cfeat.data_ptr(), w1.data_ptr(), w2.data_ptr(), gamma.data_ptr(),
            beta.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), int(ctx.training), int(ctx.relu),
            dx.data_ptr() if dx is not None else None, demb.data_ptr(), dw1.data_ptr(), db1.data_ptr(),
            dw2.data_ptr(), db2.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ctx.drop[0], ctx.drop[1], n, dim,
            ws.data_ptr(), ws.numel(), stream_ptr()), "pgnn_chem_gin_layer_bwd")
        demb = demb.view(9, dim)
        return (dx, demb[:6], demb[6:9], dw1.view(2 * dim, dim), db1, dw2.view(dim, 2 * dim), db2, dgamma, dbeta,
                None, None, None, None, None, None, None, None, None)


def chem_gin_layer(x, conv, bn, graph, relu, drop_p=0.0):
    """apply GINConv ``conv`` + BatchNorm1d ``bn`` (+ReLU, +dropout) through the fused layer call."""
    training = bn.training or bn.running_mean is None
    momentum = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            momentum = 1.0 / float(bn.num_batches_tracked)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return ChemGINLayer.apply(x, conv.edge_embedding1.weight, conv.edge_embedding2.weight, conv.mlp[0].weight,
                              conv.mlp[0].bias, conv.mlp[2].weight, conv.mlp[2].bias, bn.weight, bn.bias, graph, rm, rv,
                              training, momentum, bn.eps, relu, drop_p, dropout_seed() if drop_p > 0 else 0)


# ------------------------------------------------------------------------------------ whole chem GIN network
# Direct gradient deposit (OPT-IN).  A custom Function with 47 tensor inputs costs ~150 us per step in torch's own
# bookkeeping (input wrapping in apply(), 47 AccumulateGrad nodes; tools/autograd_floor.py: 306 vs 150 us), more
# than the library spends launching the whole forward.  With the switch on, the one-call networks take only the two
# atom embedding tables through autograd (that keeps the output attached to the graph) and write the other
# parameters' gradients into ``.grad`` themselves: assign when it is None, add otherwise -- what AccumulateGrad does.
# What this gives up: tensor hooks on those parameters and ``torch.autograd.grad(..., those parameters)`` do not see
# the gradients (torch DDP relies on such hooks; ``parallel.AllReduceOptimizers`` does not).  It is therefore OFF by
# default -- the drop-in classes keep ordinary autograd semantics -- and is switched on by ``set_direct_grads(True)``
# (bench.py and the train-step mirrors' callers do) or PGNN_DIRECT_GRADS=1.  Even when on, a network whose
# parameters carry hooks (DDP, user hooks) takes the autograd path.
_DIRECT_GRADS = os.environ.get("PGNN_DIRECT_GRADS", "0") == "1"


def set_direct_grads(on):
    """switch direct gradient deposit of the one-call networks on / off; returns the previous setting"""
    global _DIRECT_GRADS
    prev, _DIRECT_GRADS = _DIRECT_GRADS, bool(on)
    return prev


def direct_grads_enabled():
    return _DIRECT_GRADS


def _use_direct(params):
    if not (_DIRECT_GRADS and torch.is_grad_enabled()):
        return False
    for p in params:
        if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
            return False
    return True


def _private_layers(shared):
    """a backward writes gradient pointers into the layer array: work on a copy, the cached plan array is shared by
    every forward of the model (re-entrancy; two graphs of one model alive at once)"""
    return type(shared).from_buffer_copy(shared)


def _deposit_grads(params, versions, grads):
    acc_dst, acc_src = [], []
    for p, (v, ptr), g in zip(params, versions, grads):
        if p._version != v or p.data_ptr() != ptr:  # in-place update, or `.data` re-assigned (no version bump)
            raise RuntimeError("a parameter of the network was modified between forward and backward")
        if not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = g
        else:
            acc_dst.append(p.grad)
            acc_src.append(g)
    if acc_dst:
        torch._foreach_add_(acc_dst, acc_src)


class StackPlan:
    """Per-model cache of the validated, filled ``pgnn_gin_layer`` array of a one-call network.  Parameter
    storage is stable across steps (optimizers update in place), so the array is rebuilt only when a data
    pointer, a BatchNorm buffer or a momentum changes -- otherwise a step pays one tuple comparison instead
    of ~100 checks and ctypes field writes.  Keyed on data pointers, hence exact."""

    def __init__(self):
        self.key = None
        self.array = None

    def layers(self, per_layer, tensors, bns, fields):
        key = tuple(t.data_ptr() for t in tensors) + tuple(
            (b[0].data_ptr() if b[0] is not None else 0, b[1].data_ptr() if b[1] is not None else 0, b[2], b[3],
             b[4].data_ptr() if len(b) > 4 and b[4] is not None else 0) for b in bns)
        if key != self.key:
            require_cuda(*tensors)
            for t in tensors:
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise _lib.PgnnError("model parameters must be contiguous float32 tensors")
            L = (len(tensors) - 2) // per_layer
            arr = (_lib.GinLayer * L)()
            for l in range(L):
                s_, p = arr[l], tensors[2 + l * per_layer:2 + (l + 1) * per_layer]
                for name, t in zip(fields, p):
                    setattr(s_, name, t.data_ptr())
                rm, rv, momentum, eps = bns[l][:4]
                nbt = bns[l][4] if len(bns[l]) > 4 else None
                s_.running_mean = rm.data_ptr() if rm is not None else None
                s_.running_var = rv.data_ptr() if rv is not None else None
                s_.momentum, s_.eps = momentum, eps
                s_.num_batches_tracked = nbt.data_ptr() if nbt is not None else None
            self.array, self.key = arr, key
        return self.array


class ChemGINStack(Function):
    """Atom embedding + every (GINConv, BatchNorm, ReLU) layer of chem/model.py:258-277 as ONE library
    call per direction (pgnn_chem_gin_stack_fwd / _bwd).  Bit-identical to the per-layer path; valid
    for JK="last" (the pre-training and fine-tuning configurations; dropout is fused), which is when
    ``GNN.forward`` selects it.  Flat inputs: x_idx, graph, meta, xemb1, xemb2, then 8 tensors per layer
    (emb1, emb2, w1, b1, w2, b2, gamma, beta) -- see ``chem_gin_stack``."""

    PER_LAYER = 8

    @staticmethod
    def forward(ctx, x_idx, graph, meta, xemb1, xemb2, *params):
        if not x_idx.is_cuda or x_idx.dtype != torch.int64 or x_idx.dim() != 2 or x_idx.size(1) != 2:
            raise _lib.PgnnError("chem node features must be a CUDA int64 [N, 2] tensor")
        x_idx = x_idx.contiguous()
        training, bns, drop_p, drop_seed, plan, direct = meta
        if direct is not None:  # parameters travel outside autograd (see _DIRECT_GRADS)
            params = tuple(direct)
        L = len(params) // ChemGINStack.PER_LAYER
        n, dim = x_idx.size(0), xemb1.size(1)
        if training and n <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % ((n, dim),))
        dev = x_idx.device
        layers = plan.layers(8, (xemb1, xemb2) + params, bns,
                             ("emb1", "emb2", "w1", "b1", "w2", "b2", "gamma", "beta"))
        h0 = torch.empty(n, dim, dtype=torch.float32, device=dev)
        acts = torch.empty(L, 3, n, dim, dtype=torch.float32, device=dev)
        hid = torch.empty(L, n, 2 * dim, dtype=torch.float32, device=dev)
        stats = torch.empty(L, 4, dim, dtype=torch.float32, device=dev)  # mean, 1/std, scale, shift
        status = _lib.status_word(dev)
        # (the stack size, not the per-layer one: room for the bf16 planes of every layer's weights behind the op scratch)
        ws = _workspace(_ws_bytes("pgnn_chem_gin_stack_workspace_bytes", n, dim, xemb1.size(0), xemb2.size(0), L), dev)
        check(load().pgnn_chem_gin_stack_fwd(
            x_idx.data_ptr(), xemb1.data_ptr(), xemb1.size(0), xemb2.data_ptr(), xemb2.size(0), graph.in_ptr.data_ptr(),
            graph.in_src.data_ptr(), graph.in_code.data_ptr(), layers, L, int(training), h0.data_ptr(), acts.data_ptr(),
            hid.data_ptr(), stats.data_ptr(), status.data_ptr(), float(drop_p), int(drop_seed), n, dim, ws.data_ptr(),
            ws.numel(), stream_ptr()), "pgnn_chem_gin_stack_fwd")
        if _CHECK_INDICES and int(status.item()):
            raise IndexError("embedding index out of range")
        if direct is not None:
            ctx.save_for_backward(acts, hid, stats)
            ctx.direct, ctx.versions = params, [(p._version, p.data_ptr()) for p in params]
        else:
            ctx.save_for_backward(acts, hid, stats, *params)
            ctx.direct = None
        ctx.x_idx, ctx.graph, ctx.training, ctx.layers, ctx.rows = x_idx, graph, bool(training), layers, (xemb1.size(0), xemb2.size(0))
        ctx.drop = (float(drop_p), int(drop_seed))
        return acts[L - 1, 2]

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        acts, hid, stats = saved[0], saved[1], saved[2]
        L, _, n, dim = acts.shape
        dy = _rows2d(dy)
        dev = dy.device
        rows1, rows2 = ctx.rows
        sizes, shapes, byte_off = _stack_grad_layout(L, dim, rows1, rows2)
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        base = flat.data_ptr()
        layers = _private_layers(ctx.layers)
        for l in range(L):
            s, o = layers[l], byte_off[l]
            s.demb, s.dw1, s.db1, s.dw2, s.db2, s.dgamma, s.dbeta = [base + b for b in o]
        dx1, dx2 = base + byte_off[L][0], base + byte_off[L][1]
        ws = _workspace(_ws_bytes("pgnn_chem_gin_stack_workspace_bytes", n, dim, rows1, rows2, L), dev)
        g = ctx.graph
        _rows_alive = _hand_over_row_support(dy, n)  # noqa: F841
        check(load().pgnn_chem_gin_stack_bwd(
            dy.data_ptr(), dy.stride(0), ctx.x_idx.data_ptr(), rows1, rows2, g.out_ptr.data_ptr(), g.out_dst.data_ptr(),
            g.cfeat.data_ptr(), layers, L, int(ctx.training), acts.data_ptr(), hid.data_ptr(), stats.data_ptr(),
            dx1 if ctx.needs_input_grad[3] else None, dx2 if ctx.needs_input_grad[4] else None, ctx.drop[0], ctx.drop[1],
            n, dim, ws.data_ptr(), ws.numel(), stream_ptr()), "pgnn_chem_gin_stack_bwd")
        pieces = flat.split_with_sizes(sizes)
        grads = tuple(t if shp is None else t.view(shp) for t, shp in zip(pieces, shapes))
        if ctx.direct is not None:
            _deposit_grads(ctx.direct, ctx.versions, grads[2:])
            return (None, None, None) + grads[:2]
        global _autograd_path_backwards
        _autograd_path_backwards += 1  # these gradients reach .grad through AccumulateGrad, behind the whole backward (parallel.py)
        return (None, None, None) + grads


_autograd_path_backwards = 0


def autograd_path_backwards():
    """how many chem GIN stack backwards handed their parameter gradients to autograd instead of depositing them (hooks on a
    parameter, direct deposit off): what the overlapped all-reduce compares before trusting a gradient milestone"""
    return _autograd_path_backwards


_stack_layouts = {}


def _stack_grad_layout(L, dim, rows1, rows2):
    """one flat gradient buffer for the stack backward: sizes / 2-D shapes of its pieces in the order
    ChemGINStack.forward takes the parameters, and the byte offsets the C structs point at."""
    key = (L, dim, rows1, rows2)
    lay = _stack_layouts.get(key)
    if lay is None:
        sizes = [rows1 * dim, rows2 * dim]
        shapes = [(rows1, dim), (rows2, dim)]
        byte_off = []
        for _ in range(L):
            off = sum(sizes) * 4
            # emb1 [6,d] and emb2 [3,d] are adjacent: together they are the C side's demb [9,d]
            per = [(6 * dim, (6, dim)), (3 * dim, (3, dim)), (2 * dim * dim, (2 * dim, dim)), (2 * dim, None),
                   (2 * dim * dim, (dim, 2 * dim)), (dim, None), (dim, None), (dim, None)]
            starts, o = [], off
            for sz, _shp in per:
                starts.append(o)
                o += sz * 4
            byte_off.append([starts[0]] + starts[2:])  # demb, dw1, db1, dw2, db2, dgamma, dbeta
            sizes += [sz for sz, _ in per]
            shapes += [shp for _, shp in per]
        byte_off.append([0, rows1 * dim * 4])
        lay = _stack_layouts[key] = (sizes, shapes, byte_off)
    return lay


_plans = weakref.WeakKeyDictionary()  # GNN module -> StackPlan (kept off the module: ctypes arrays do not deepcopy)


def _bn_meta(bns, in_call=False):
    """(running_mean, running_var, momentum, eps, counter) per layer, as nn.BatchNorm1d.forward would handle them.
    ``num_batches_tracked`` of a training-mode forward: incremented here (one torch launch for all layers, ~70 us of host
    time), or -- ``in_call``: the GIN stack calls -- handed to the library as the fifth entry, which increments it inside a
    launch it makes anyway.  momentum=None (cumulative average) needs the count on the host first and stays on the torch path."""
    meta, counters = [], []
    for bn in bns:
        momentum = 0.0 if bn.momentum is None else bn.momentum
        counter = None
        buf = bn._buffers  # (running_mean / running_var / num_batches_tracked without three trips through nn.Module.__getattr__)
        nbt = buf.get("num_batches_tracked")
        if bn.training and bn.track_running_stats and nbt is not None:
            if bn.momentum is None:
                momentum = 1.0 / float(nbt + 1)
                counters.append(nbt)
            elif in_call and nbt.is_cuda and nbt.dtype == torch.int64:
                counter = nbt
            else:
                counters.append(nbt)
        meta.append((buf.get("running_mean") if bn.track_running_stats else None,
                     buf.get("running_var") if bn.track_running_stats else None, float(momentum), float(bn.eps), counter))
    if counters:
        torch._foreach_add_(counters, 1)  # num_batches_tracked of every layer in one launch
    return meta


def _chem_gin_flat(plan, convs, bns):
    """the 8 parameters per layer of a chem GIN network, in StackPlan order.  Forty ``conv.mlp[0].weight``-style reads go through
    nn.Module.__getattr__ / Sequential.__getitem__ (~1 us each, 45 us per call: the host-bound context-prediction step makes four
    such calls); the sub-MODULES are remembered instead, re-checked by identity through the ``_modules`` dicts, and the parameters
    read from their ``_parameters`` dicts every call (a replaced Parameter or sub-module is picked up)."""
    cache = plan.__dict__.get("flat_mods")
    cm, bm = convs._modules, bns._modules
    if cache is not None and len(cache) == len(cm) == len(bm):
        flat = []
        for (conv, e1, e2, mlp, l0, l2, bn), c, b in zip(cache, cm.values(), bm.values()):
            m = c._modules
            if c is not conv or b is not bn or m.get("edge_embedding1") is not e1 or m.get("edge_embedding2") is not e2 or m.get("mlp") is not mlp:
                break
            mm = mlp._modules
            if mm.get("0") is not l0 or mm.get("2") is not l2:
                break
            p0, p2, pb = l0._parameters, l2._parameters, bn._parameters
            flat += (e1._parameters["weight"], e2._parameters["weight"], p0["weight"], p0["bias"], p2["weight"], p2["bias"],
                     pb["weight"], pb["bias"])
        else:
            return flat
    plan.flat_mods = [(conv, conv.edge_embedding1, conv.edge_embedding2, conv.mlp, conv.mlp[0], conv.mlp[2], bn)
                      for conv, bn in zip(convs, bns)]
    return [t for conv, bn in zip(convs, bns) for t in (
        conv.edge_embedding1.weight, conv.edge_embedding2.weight, conv.mlp[0].weight, conv.mlp[0].bias,
        conv.mlp[2].weight, conv.mlp[2].bias, bn.weight, bn.bias)]


def chem_gin_stack(owner, x_idx, graph, x_embedding1, x_embedding2, convs, bns, drop_p=0.0):
    """run the atom embedding and all (conv, bn) layers through the stack call; ReLU after every layer
    but the last and (``drop_p`` > 0) dropout after every layer, as GNN.forward of the reference does.
    ``owner`` (the GNN module) keys the cached StackPlan."""
    plan = _plans.get(owner)
    if plan is None:
        plan = _plans[owner] = StackPlan()
    training = bns[0].training or bns[0].running_mean is None
    flat = _chem_gin_flat(plan, convs, bns)
    seed = dropout_seed() if drop_p > 0 else 0
    if x_embedding1.weight.requires_grad and _use_direct(flat):
        return ChemGINStack.apply(x_idx, graph, (training, _bn_meta(bns, in_call=True), drop_p, seed, plan, flat),
                                  x_embedding1.weight, x_embedding2.weight)
    return ChemGINStack.apply(x_idx, graph, (training, _bn_meta(bns, in_call=True), drop_p, seed, plan, None),
                              x_embedding1.weight, x_embedding2.weight, *flat)


# ------------------------------------------------------------------------------------ products on pre-split weight planes
def weight_planes(mats, transpose=None):
    """pgnn_split_weights: the three-term bf16 split of every fp32 matrix in ``mats`` (``transpose[j]``: of its transpose), one
    launch; returns one int16 tensor [3, rows, ld] per matrix (ld = columns rounded up to 32, zero padded) -- what
    ``linear_fwd_wp`` / pgnn_linear_bwd_data_wp take as their weight operand"""
    import ctypes
    cnt = len(mats)
    transpose = [False] * cnt if transpose is None else list(transpose)
    outs = []
    for w, tr in zip(mats, transpose):
        if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or not w.is_contiguous():
            raise _lib.PgnnError("weight_planes: contiguous fp32 CUDA matrices only")
        r, c = (w.size(1), w.size(0)) if tr else (w.size(0), w.size(1))
        outs.append(torch.empty(3, r, (c + 31) // 32 * 32, dtype=torch.int16, device=w.device))
    arr = lambda vals, ty: (ty * cnt)(*vals)
    check(load().pgnn_split_weights(arr([w.data_ptr() for w in mats], ctypes.c_void_p), arr([o.data_ptr() for o in outs], ctypes.c_void_p),
                                    arr([w.size(0) for w in mats], ctypes.c_int64), arr([w.size(1) for w in mats], ctypes.c_int64),
                                    arr([int(t) for t in transpose], ctypes.c_int32), cnt, stream_ptr()), "pgnn_split_weights")
    return outs


def linear_fwd_wp(x, planes, bias, n_out, relu=False, out=None):
    """y = act(x . W^T + b) with W given as its planes (``weight_planes([W])[0]``): pgnn_linear_fwd_wp, no autograd"""
    m, k = x.shape
    y = out if out is not None else torch.empty(m, n_out, dtype=torch.float32, device=x.device)
    check(load().pgnn_linear_fwd_wp(x.data_ptr(), x.stride(0), planes.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                    y.stride(0), m, k, n_out, int(relu), None, stream_ptr()), "pgnn_linear_fwd_wp")
    return y


def weight_planes_2p(mats, transpose=None):
    """pgnn_split_weights_2p: every fp32 matrix of ``mats`` (``transpose[j]``: its transpose) as two fp16 planes of s W with a
    power-of-two scale s per row, followed by 1 / s per row as fp32 -- one uint8 buffer of pgnn_weight_planes_bytes per matrix: what
    ``linear_fwd_2p`` / pgnn_linear_bwd_data_2p take as their weight operand (the default arithmetic of the one-call networks)"""
    import ctypes
    cnt = len(mats)
    transpose = [False] * cnt if transpose is None else list(transpose)
    outs = []
    for w, tr in zip(mats, transpose):
        if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or not w.is_contiguous():
            raise _lib.PgnnError("weight_planes_2p: contiguous fp32 CUDA matrices only")
        r, c = (w.size(1), w.size(0)) if tr else (w.size(0), w.size(1))
        outs.append(torch.empty(int(load().pgnn_weight_planes_bytes(r, c)), dtype=torch.uint8, device=w.device))
    arr = lambda vals, ty: (ty * cnt)(*vals)
    check(load().pgnn_split_weights_2p(arr([w.data_ptr() for w in mats], ctypes.c_void_p), arr([o.data_ptr() for o in outs], ctypes.c_void_p),
                                       arr([w.size(0) for w in mats], ctypes.c_int64), arr([w.size(1) for w in mats], ctypes.c_int64),
                                       arr([int(t) for t in transpose], ctypes.c_int32), cnt, stream_ptr()), "pgnn_split_weights_2p")
    return outs


def linear_fwd_2p(x, planes, bias, n_out, relu=False, out=None, x_amax=None, y_amax=None):
    """y = act(x . W^T + b) with W as ``weight_planes_2p([W])[0]``: pgnn_linear_fwd_2p, no autograd.  x_amax: int32 [m] bit patterns
    of the rows' largest magnitudes if a previous product left them (its y_amax), else the kernel takes them; y_amax: zeroed int32
    [m] that receives the result rows' maxima"""
    m, k = x.shape
    y = out if out is not None else torch.empty(m, n_out, dtype=torch.float32, device=x.device)
    check(load().pgnn_linear_fwd_2p(x.data_ptr(), x.stride(0), x_amax.data_ptr() if x_amax is not None else None, planes.data_ptr(),
                                    bias.data_ptr() if bias is not None else None, y.data_ptr(), y.stride(0), m, k, n_out, int(relu), None,
                                    y_amax.data_ptr() if y_amax is not None else None, stream_ptr()), "pgnn_linear_fwd_2p")
    return y


# ------------------------------------------------------------------------------------ whole bio GIN network
class BioGINStack(Function):
    """Every (GINConv, ReLU) layer of the bio GNN (bio/model.py:11-58, 258-290, JK = "last", no dropout) as ONE library call
    per direction (pgnn_bio_gin_stack_fwd / _bwd): graph-resident concat aggregation, both products of the mlp, the
    BatchNorm1d(2D) between them, the ReLU between layers fused into the second product's epilogue; backward with the
    weight-gradient products on the side stream and backward-data on transposed weights.  Inputs: h0 [N, D] (layer 0's
    embedded input), graph, meta, then 8 tensors per layer (enc_w [D,9], enc_b [D], w1 [2D,2D], b1 [2D], w2 [D,2D], b2 [D],
    gamma [2D], beta [2D]) -- see ``bio_gin_stack``."""

    PER_LAYER = 8

    @staticmethod
    def forward(ctx, h0, graph, meta, *params):
        training, bns = meta
        require_cuda(h0, *params)
        h0 = _rows2d(h0)
        n, dim = h0.shape
        L = len(params) // BioGINStack.PER_LAYER
        if graph.kind != "bio" or graph.gcn or n != graph.n:
            raise _lib.PgnnError("bio GIN stack: needs the bio graph built with gcn=False for these nodes")
        if training and n <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % ((n, 2 * dim),))
        dev = h0.device
        for t in params:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.PgnnError("model parameters must be contiguous float32 tensors")
        layers = (_lib.GinLayer * L)()
        for l in range(L):
            s_, p = layers[l], params[8 * l:8 * l + 8]
            # edge_encoder.weight [D, 9] / .bias [D] as they are: the call writes the [10, D] tables [W^T; b] in the launch that
            # splits the weights (three torch.cat / stack launches per step otherwise)
            s_.emb1, s_.emb2 = p[0].data_ptr(), p[1].data_ptr()
            s_.w1, s_.b1, s_.w2, s_.b2, s_.gamma, s_.beta = [t.data_ptr() for t in p[2:]]
            rm, rv, momentum, eps, nbt = bns[l]
            s_.running_mean = rm.data_ptr() if rm is not None else None
            s_.running_var = rv.data_ptr() if rv is not None else None
            s_.momentum, s_.eps = momentum, eps
            s_.num_batches_tracked = nbt.data_ptr() if nbt is not None else None
        acts = torch.empty(L, 7, n, dim, dtype=torch.float32, device=dev)
        stats = torch.empty(L, 2, 2 * dim, dtype=torch.float32, device=dev)
        ws = _workspace(_ws_bytes("pgnn_bio_gin_stack_workspace_bytes", n, dim, L), dev)
        tiles = graph.tiles
        check(load().pgnn_bio_gin_stack_fwd(
            h0.data_ptr(), h0.stride(0), graph.in_ptr.data_ptr(), graph.in_src.data_ptr(), graph.cfeat.data_ptr(),
            tiles[0].data_ptr() if tiles is not None else None, tiles[1].data_ptr() if tiles is not None else None, layers, L,
            int(training), acts.data_ptr(), stats.data_ptr(), n, dim, ws.data_ptr(), ws.numel(), stream_ptr()), "pgnn_bio_gin_stack_fwd")
        ctx.save_for_backward(acts, stats, *params)
        ctx.graph, ctx.training, ctx.layers = graph, bool(training), layers
        return acts[L - 1, 6]

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        acts, stats = ctx.saved_tensors[:2]
        L, _, n, dim = acts.shape
        dy = _rows2d(dy)
        dev = dy.device
        graph = ctx.graph
        # one flat gradient buffer: per layer d enc_w [D, 9] + d enc_b [D] (one piece for the call: the layers carry emb2, so the
        # gradient comes in the module's layout and autograd takes the views as they are -- a [9, D] result transposed here cost
        # one copy launch per layer), dw1 [2D, 2D], db1 [2D], dw2 [D, 2D], db2 [D], dgamma [2D], dbeta [2D]
        per = [10 * dim, 4 * dim * dim, 2 * dim, 2 * dim * dim, dim, 2 * dim, 2 * dim]
        flat = torch.empty(L * sum(per), dtype=torch.float32, device=dev)
        layers = _private_layers(ctx.layers)
        views = []
        for l in range(L):
            pieces = flat[l * sum(per):(l + 1) * sum(per)].split_with_sizes(per)
            s_ = layers[l]
            s_.demb, s_.dw1, s_.db1, s_.dw2, s_.db2, s_.dgamma, s_.dbeta = [t.data_ptr() for t in pieces]
            views.append(pieces)
        dh0 = torch.empty(n, dim, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        ws = _workspace(_ws_bytes("pgnn_bio_gin_stack_workspace_bytes", n, dim, L), dev)
        tiles = graph.tiles
        check(load().pgnn_bio_gin_stack_bwd(
            dy.data_ptr(), dy.stride(0), graph.out_ptr.data_ptr(), graph.out_dst.data_ptr(), graph.cfeat.data_ptr(),
            tiles[0].data_ptr() if tiles is not None else None, tiles[1].data_ptr() if tiles is not None else None, layers, L,
            int(ctx.training), acts.data_ptr(), stats.data_ptr(), dh0.data_ptr() if dh0 is not None else None, n, dim, ws.data_ptr(),
            ws.numel(), stream_ptr()), "pgnn_bio_gin_stack_bwd")
        grads = []
        for l in range(L):
            denc, dw1, db1, dw2, db2, dgam, dbet = views[l]
            grads += [denc[:9 * dim].view(dim, 9), denc[9 * dim:], dw1.view(2 * dim, 2 * dim), db1, dw2.view(dim, 2 * dim), db2, dgam, dbet]
        return (dh0, None, None) + tuple(grads)


def bio_gin_stack(h0, graph, convs):
    """all GINConv layers of the bio GNN (ReLU between them) through the stack call; ``h0`` = layer 0's embedded input"""
    bns = [c.mlp[1] for c in convs]
    training = bns[0].training or bns[0].running_mean is None
    flat = [t for c in convs for t in (c.edge_encoder.weight, c.edge_encoder.bias, c.mlp[0].weight, c.mlp[0].bias,
                                       c.mlp[3].weight, c.mlp[3].bias, c.mlp[1].weight, c.mlp[1].bias)]
    return BioGINStack.apply(h0, graph, (training, _bn_meta(bns, in_call=True)), *flat)


# ------------------------------------------------------------------------------------ whole chem GCN / GraphSAGE network
_lin_layouts = {}


def _lin_grad_layout(L, dim, rows1, rows2):
    key = (L, dim, rows1, rows2)
    lay = _lin_layouts.get(key)
    if lay is None:
        sizes, shapes, byte_off = [rows1 * dim, rows2 * dim], [(rows1, dim), (rows2, dim)], []
        per = [(6 * dim, (6, dim)), (3 * dim, (3, dim)), (dim * dim, (dim, dim)), (dim, None), (dim, None), (dim, None)]
        for _ in range(L):
            o, starts = sum(sizes) * 4, []
            for sz, _shp in per:
                starts.append(o)
                o += sz * 4
            byte_off.append([starts[0]] + starts[2:])  # demb (emb1+emb2 adjacent), dw, db, dgamma, dbeta
            sizes += [sz for sz, _ in per]
            shapes += [shp for _, shp in per]
        byte_off.append([0, rows1 * dim * 4])
        lay = _lin_layouts[key] = (sizes, shapes, byte_off)
    return lay


class ChemLinStack(Function):
    """Atom embedding + every (GCNConv | GraphSAGEConv, BatchNorm, ReLU, dropout) layer of chem/model.py as ONE
    library call per direction (pgnn_chem_lin_stack_fwd / _bwd, kind 1 = GCN, 2 = GraphSAGE).  Same kernels and
    order as the per-layer path (bit-identical); JK="last".  Flat inputs: x_idx, graph, meta, xemb1, xemb2, then
    6 tensors per layer (emb1, emb2, w, b, gamma, beta)."""

    PER_LAYER = 6

    @staticmethod
    def forward(ctx, x_idx, graph, meta, xemb1, xemb2, *params):
        if not x_idx.is_cuda or x_idx.dtype != torch.int64 or x_idx.dim() != 2 or x_idx.size(1) != 2:
            raise _lib.PgnnError("chem node features must be a CUDA int64 [N, 2] tensor")
        x_idx = x_idx.contiguous()
        kind, training, bns, drop_p, drop_seed, plan, direct = meta
        if direct is not None:
            params = tuple(direct)
        L = len(params) // ChemLinStack.PER_LAYER
        n, dim = x_idx.size(0), xemb1.size(1)
        if training and n <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % ((n, dim),))
        if (kind == 1) != bool(graph.gcn):
            raise _lib.PgnnError("graph structure was built for the other convolution type")
        dev = x_idx.device
        layers = plan.layers(6, (xemb1, xemb2) + params, bns, ("emb1", "emb2", "w1", "b1", "gamma", "beta"))
        h0 = torch.empty(n, dim, dtype=torch.float32, device=dev)
        acts = torch.empty(L, 4, n, dim, dtype=torch.float32, device=dev)  # lin, sum, z, y
        norms = torch.empty(L, n, dtype=torch.float32, device=dev) if kind == 2 else None
        stats = torch.empty(L, 4, dim, dtype=torch.float32, device=dev)
        status = _lib.status_word(dev)
        ws = _workspace(_ws_bytes("pgnn_chem_gin_layer_workspace_bytes", n, dim), dev)
        check(load().pgnn_chem_lin_stack_fwd(
            kind, x_idx.data_ptr(), xemb1.data_ptr(), xemb1.size(0), xemb2.data_ptr(), xemb2.size(0),
            graph.in_ptr.data_ptr(), graph.in_src.data_ptr(), graph.in_code.data_ptr(),
            graph.dinv.data_ptr() if kind == 1 else None, layers, L, int(training), h0.data_ptr(), acts.data_ptr(),
            norms.data_ptr() if norms is not None else None, stats.data_ptr(), status.data_ptr(), float(drop_p),
            int(drop_seed), n, dim, ws.data_ptr(), ws.numel(), stream_ptr()), "pgnn_chem_lin_stack_fwd")
        if _CHECK_INDICES and int(status.item()):
            raise IndexError("embedding index out of range")
        saved = [h0, acts, stats] + ([norms] if norms is not None else [])
        if direct is not None:
            ctx.direct, ctx.versions = params, [(p._version, p.data_ptr()) for p in params]
        else:
            saved += list(params)
            ctx.direct = None
        ctx.save_for_backward(*saved)
        ctx.kind, ctx.x_idx, ctx.graph, ctx.training, ctx.layers = kind, x_idx, graph, bool(training), layers
        ctx.rows, ctx.drop = (xemb1.size(0), xemb2.size(0)), (float(drop_p), int(drop_seed))
        return acts[L - 1, 3]

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        h0, acts, stats = saved[0], saved[1], saved[2]
        norms = saved[3] if ctx.kind == 2 else None
        L, _, n, dim = acts.shape
        dy = _rows2d(dy)
        dev = dy.device
        rows1, rows2 = ctx.rows
        sizes, shapes, byte_off = _lin_grad_layout(L, dim, rows1, rows2)
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        base = flat.data_ptr()
        layers = _private_layers(ctx.layers)
        for l in range(L):
            s, o = layers[l], byte_off[l]
            s.demb, s.dw1, s.db1, s.dgamma, s.dbeta = [base + b for b in o]
        dx1, dx2 = base + byte_off[L][0], base + byte_off[L][1]
        ws = _workspace(_ws_bytes("pgnn_chem_lin_stack_workspace_bytes", n, dim, rows1, rows2), dev)
        g = ctx.graph
        check(load().pgnn_chem_lin_stack_bwd(
            ctx.kind, dy.data_ptr(), dy.stride(0), ctx.x_idx.data_ptr(), rows1, rows2, g.in_ptr.data_ptr(),
            g.out_ptr.data_ptr(), g.out_dst.data_ptr(), g.dinv.data_ptr() if ctx.kind == 1 else None, g.cfeat.data_ptr(),
            layers, L, int(ctx.training), h0.data_ptr(), acts.data_ptr(), norms.data_ptr() if norms is not None else None,
            stats.data_ptr(), dx1 if ctx.needs_input_grad[3] else None, dx2 if ctx.needs_input_grad[4] else None,
            ctx.drop[0], ctx.drop[1], n, dim, ws.data_ptr(), ws.numel(), stream_ptr()), "pgnn_chem_lin_stack_bwd")
        pieces = flat.split_with_sizes(sizes)
        grads = tuple(t if shp is None else t.view(shp) for t, shp in zip(pieces, shapes))
        if ctx.direct is not None:
            _deposit_grads(ctx.direct, ctx.versions, grads[2:])
            return (None, None, None) + grads[:2]
        return (None, None, None) + grads


def chem_lin_stack(owner, kind, x_idx, graph, x_embedding1, x_embedding2, convs, bns, drop_p=0.0):
    """GCN (kind 1) / GraphSAGE (kind 2) network through the one-call path; see ``chem_gin_stack``."""
    plan = _plans.get(owner)
    if plan is None:
        plan = _plans[owner] = StackPlan()
    training = bns[0].training or bns[0].running_mean is None
    flat = [t for conv, bn in zip(convs, bns) for t in (
        conv.edge_embedding1.weight, conv.edge_embedding2.weight, conv.linear.weight, conv.linear.bias, bn.weight, bn.bias)]
    seed = dropout_seed() if drop_p > 0 else 0
    if x_embedding1.weight.requires_grad and _use_direct(flat):
        return ChemLinStack.apply(x_idx, graph, (kind, training, _bn_meta(bns), drop_p, seed, plan, flat),
                                  x_embedding1.weight, x_embedding2.weight)
    return ChemLinStack.apply(x_idx, graph, (kind, training, _bn_meta(bns), drop_p, seed, plan, None),
                              x_embedding1.weight, x_embedding2.weight, *flat)
